#!/usr/bin/env python3
"""Which hardware queues do the kernels of a rocprofv3 --kernel-trace csv run on?   python tools/probes/trace_queues.py <dir>"""
import collections, csv, glob, sys
q = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        q[(r.get("Queue_Id"), r.get("Stream_Id", ""))][r["Kernel_Name"].split("(")[0][:50]] += 1
for k, c in q.items():
    print("queue %s stream %s: %d kernels; top: %s" % (k[0], k[1], sum(c.values()), ", ".join("%s x%d" % kv for kv in c.most_common(6))))
