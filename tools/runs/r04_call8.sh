#!/bin/bash
# round 4, GPU call 8: thread order of the F(4,3) operand writer (A/B through I2V_MOD_ORDER)
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
for ord in 0 1; do
  I2V_MOD_ORDER=$ord timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64_ord$ord.json
  I2V_MOD_ORDER=$ord timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128_ord$ord.json
  I2V_MOD_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bair$ord -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bair$ord.log 2>&1
  I2V_MOD_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_land$ord -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_land$ord.log 2>&1
done
rm -f $O/prof_*/bench_kernel_trace.csv
timeout 200 python -m pytest tests -m gpu -x -q -k "structure_switches or nf8_bair or full_width_bair" 2>&1 | tail -2
python3 - <<'PY'
import json, csv
for o in (0, 1):
    for f in ("bench_bair64", "bench_land128"):
        d = json.load(open(f"gpurun_out/r04h/{f}_ord{o}.json")); print(f, "order", o, round(d["ms_per_step"], 3), "ms")
    for t in ("bair", "land"):
        for r in csv.DictReader(open(f"gpurun_out/r04h/prof_{t}{o}/bench_kernel_stats.csv")):
            if "modulate_wino4" in r["Name"] or "conv_wino4_f16x3_kernel<9" in r["Name"]: print("  ", t, "order", o, r["Name"][:55], r["Calls"], round(float(r["TotalDurationNs"]) / 5e6, 3), "ms per pass")
PY
