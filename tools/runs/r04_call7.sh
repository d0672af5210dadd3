#!/bin/bash
# round 4, GPU call 7: operand writer with 4 channels (hi and lo) per thread, full suite, bench
export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^$|amdgpu.ids" > $O/pytest.log; echo "pytest rc ${PIPESTATUS[0]}" >> $O/pytest.log )
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bair -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bair.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_land -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_land.log 2>&1
rm -f $O/prof_*/bench_kernel_trace.csv
tail -3 $O/pytest.log
python3 - <<'PY'
import json, csv
for f in ("bench_bair64", "bench_land128"):
    d = json.load(open(f"gpurun_out/r04g/{f}.json")); print(f, round(d["ms_per_step"], 3), "ms")
for t in ("bair", "land"):
    for r in csv.DictReader(open(f"gpurun_out/r04g/prof_{t}/bench_kernel_stats.csv")):
        if "modulate" in r["Name"]: print(t, r["Name"][:60], r["Calls"], round(float(r["TotalDurationNs"]) / 5e6, 3), "ms per pass")
PY
