"""Coordinate search over the ride-along share weights (FLOAT_DEC_RIDE_W, rows 64 / 128 / 256 / 512 px x up-conv, conv2, flow): each
candidate is one run of dec_host2.py in a fresh process (the weights are read once per process)."""
import os, re, subprocess, sys
base = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "170 140 50 170 155 100 220 200 180 300 250 120").split()]


def run(w, reps=1):
    best = 1e9
    for _ in range(reps):
        env = dict(os.environ, FLOAT_DEC_RIDE_W=" ".join("%.0f" % x for x in w))
        out = subprocess.run([sys.executable, "tools/probes/dec_host2.py"], env=env, capture_output=True, text=True).stdout
        m = re.search(r"([\d.]+) ms", out)
        best = min(best, float(m.group(1)) if m else 1e9)
    return best


cur = run(base, 2)
print("start %s -> %.2f ms" % (base, cur), flush=True)
for sweep in range(2):
    for i in range(12):
        for f in (0.65, 1.5):
            cand = list(base)
            cand[i] = max(10.0, cand[i] * f)
            t = run(cand)
            if t < cur - 0.08:
                t = max(t, run(cand))  # confirm
                if t < cur - 0.05:
                    base, cur = cand, t
                    print("  weight %d x%.2f -> %.2f ms  %s" % (i, f, cur, " ".join("%.0f" % x for x in base)), flush=True)
                    break
print("best %.2f ms: %s" % (cur, " ".join("%.0f" % x for x in base)))
