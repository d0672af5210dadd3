"""Idle gaps of the main queue inside ONE steady-state step of the pipelined bench loop (run on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --steps 4 --warmup 2 --lean
    python tools/step_gaps.py <dir> [min_gap_us]
The step = the window between the ends of the last two conv_img launches (one per decoder pass).  Prints, per hardware queue, the summed
kernel time inside the window, the main queue's busy fraction, and every gap of the main queue above min_gap_us with the launches on
either side -- event waits for side-stream work, launch-bound stretches and tails show up here."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 1 else "Queue_Id"


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("i2v::", "")[:60]


ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[0])
ci = [k for k in ks if "conv_img" in k[3]]
if len(ci) < 2:
    raise SystemExit("fewer than two conv_img launches in the trace")
w0, w1, mainq = ci[-2][1], ci[-1][1], ci[-1][2]
win = [k for k in ks if k[1] > w0 and k[0] < w1]
print(f"step window {(w1 - w0) / 1e6:.3f} ms, {len(win)} launches, queues by {qkey}; main queue {mainq}")
busy = defaultdict(float)
cnt = defaultdict(int)
for s, e, q, n, _ in win:
    busy[q] += (min(e, w1) - max(s, w0)) / 1e3
    cnt[q] += 1
for q in sorted(busy, key=lambda q: -busy[q]):
    print(f"  queue {q:>4s}: {cnt[q]:4d} launches, {busy[q] / 1e3:8.3f} ms of kernel time ({100 * busy[q] * 1e3 / (w1 - w0):5.1f} % of the window)")
main = [k for k in win if k[2] == mainq]
tot_gap, big = 0.0, []
for a, b in zip(main, main[1:]):
    g = (b[0] - a[1]) / 1e3
    if g > 0:
        tot_gap += g
    if g >= min_gap:
        big.append((g, a[3], b[3], (a[1] - w0) / 1e6))
print(f"main queue: {len(main)} launches, idle between launches {tot_gap / 1e3:.3f} ms in total; gaps >= {min_gap} us:")
for g, a, b, t in big:
    print(f"  at {t:7.3f} ms  {g:8.1f} us   {a}  ->  {b}")
# what ran on the other queues during the main queue's big gaps
for g, a, b, t in sorted(big, reverse=True)[:5]:
    lo = w0 + t * 1e6
    hi = lo + g * 1e3
    oth = [k for k in win if k[2] != mainq and k[1] > lo and k[0] < hi]
    print(f"  during the {g:.0f} us gap at {t:.3f} ms: " + (", ".join(sorted({k[3][:40] for k in oth})) or "nothing on any queue"))

qmap = defaultdict(set)
for k in ks:
    qmap[k[2]].add(k[4])
print("stream -> hardware queue ids: " + "; ".join(f"{a} -> {sorted(b)}" for a, b in sorted(qmap.items())))
# optional dump of every launch in [t0, t1] ms of the window: python tools/step_gaps.py <dir> <min_gap> <t0> <t1>
if len(sys.argv) > 4:
    t0, t1 = float(sys.argv[3]), float(sys.argv[4])
    print(f"launches between {t0} and {t1} ms of the window (start ms, duration us, stream, queue, kernel):")
    for k in win:
        a = (k[0] - w0) / 1e6
        if t0 <= a <= t1 or t0 <= (k[1] - w0) / 1e6 <= t1:
            print(f"  {a:8.3f}  {(k[1] - k[0]) / 1e3:8.1f}  s{k[2]:>2s} q{k[4]:>2s}  {k[3]}")
