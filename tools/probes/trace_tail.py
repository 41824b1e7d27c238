"""Diagnostic: the last launches of one decode + hand-over under `rocprofv3 --kernel-trace --memory-copy-trace`: what runs after
the final flow kernel (the exposed copy of the last batch) and how long it takes.  python tools/probes/trace_tail.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:50], r.get('Grid_Size_X', r.get('Grid_Size', ''))))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'MEMCPY ' + r.get('Direction', '') , r.get('Bytes', r.get('Size', ''))))
rows.sort()
t0 = rows[-60][0] if len(rows) > 60 else rows[0][0]
for s, e, n, g in rows[-60:]:
    print("%9.1f +%8.1f us  %-52s %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, g))
