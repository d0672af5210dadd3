#!/bin/bash
# round 4, GPU call 12: PIPE = 2 ("lite") and PIPE = 1 after the register fix, against PIPE = 0
export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "structure_switches" 2>&1 | tail -2
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 256 128 1 0" "8 16 64 64 64 64 0 1" "8 16 64 64 128 64 1 0" "8 8 32 32 256 256 0 1" "4 16 128 128 64 32 0 0" "4 16 128 128 32 32 0 1" "8 4 16 16 512 512 0 1" "32 16 64 64 128 128 0 1"; do
  for pipe in 0 1 2; do
    echo "== PIPE $pipe shape $s" >> $O/pipe_ab.txt
    I2V_W4_PIPE=$pipe timeout 120 tools/conv16w_check $s 2>&1 | grep -E "F\(4,3\)|fault|error|Error" >> $O/pipe_ab.txt
  done
done
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 64 64 0 1" "4 16 128 128 32 32 0 1"; do
  for pipe in 0 2; do
    echo "== PIPE $pipe" >> $O/f43_timeline.txt
    I2V_W4_PIPE=$pipe timeout 120 tools/conv16w_check_tl $s 2>&1 | grep -v "^$" | grep -vE "direct |F\(2,3\) |wino vs" >> $O/f43_timeline.txt
  done
done
for pipe in 0 2 1; do
  I2V_W4_PIPE=$pipe timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64_pipe$pipe.json
  I2V_W4_PIPE=$pipe timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128_pipe$pipe.json
done
grep -E "PIPE|ms " $O/pipe_ab.txt | paste - - | awk '{print $3, $5,$6,$7,$8,$9,$10,$11,$12, $(NF-4), $(NF-3)}'
python3 - <<'PY'
import json
for p in (0, 2, 1):
    for f in ("bench_bair64", "bench_land128"):
        d = json.load(open(f"gpurun_out/r04l/{f}_pipe{p}.json")); print(f, "pipe", p, round(d["ms_per_step"], 3), "ms; F(4,3)", round(d["roofline"]["ms_per_step"], 2))
PY
