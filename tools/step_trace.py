"""Per-launch durations of the decoder kernels of ONE bench step (run on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras
    python tools/step_trace.py <dir> [min_us]"""
import csv, glob, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("i2v::", "")
    if "rocclr" in nm or nm.startswith("at::"):
        continue
    seq.append((nm[:44], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
dec = [s for s in seq if not s[0].startswith("flow_")]
half = dec[len(dec) // 2:]          # the timed step (the first half is the warm-up step)
tot = 0.0
for nm, us in half:
    tot += us
    if us >= min_us:
        print(f"{nm:46s} {us:9.1f} us")
print(f"decoder kernels of the step: {len(half)} launches, {tot / 1e3:.2f} ms")
