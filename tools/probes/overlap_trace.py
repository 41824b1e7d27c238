"""Why does FMT(k+1) || decode(k) lose?  From a rocprofv3 --kernel-trace of `bench.py --quick --steps 3 [--overlap prio]`:
the chain's kernels (fmt_*) of the LAST clip - mean duration and mean start-to-start distance - split into the stretch where no
decoder kernel is in flight and the stretch beside decoder kernels.
    python tools/probes/overlap_trace.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
fmt = [r for r in rows if "fmt_" in r[2]]
dec = [r for r in rows if "dec_" in r[2]]
# the last clip: the last 250 * 59 chain launches + 5 adaLN launches
fmt = fmt[-(250 * 59 + 10):]
t0 = fmt[0][0]
dec = [r for r in dec if r[1] > t0]
import bisect
ds, de = [r[0] for r in dec], [r[1] for r in dec]
def beside_decoder(s, e):
    i = bisect.bisect_right(ds, e)
    return any(de[j] > s for j in range(max(0, i - 8), i))
stats = {False: [0, 0.0, 0.0], True: [0, 0.0, 0.0]}
for a, b in zip(fmt, fmt[1:]):
    k = beside_decoder(a[0], a[1])
    st = stats[k]
    st[0] += 1
    st[1] += (a[1] - a[0]) / 1e3
    st[2] += (b[0] - a[0]) / 1e3
print("queues: chain %s, decoder %s" % (sorted(set(r[3] for r in fmt)), sorted(set(r[3] for r in dec))))
print("last clip: chain span %.1f ms (%d launches), decoder kernels in it: %d, clip span %.1f ms" % (
    (fmt[-1][1] - fmt[0][0]) / 1e6, len(fmt), len(dec), (max(fmt[-1][1], dec[-1][1] if dec else 0) - fmt[0][0]) / 1e6))
for k in (False, True):
    n, du, ss = stats[k]
    if n:
        print("chain launches %s a decoder kernel: %6d, mean duration %.2f us, mean start-to-start %.2f us" % ("BESIDE" if k else "without", n, du / n, ss / n))
