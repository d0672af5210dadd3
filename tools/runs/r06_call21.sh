#!/bin/bash
# round 6, GPU call 21: the ~1 ms idle stretch of the main stream at the top of every step: which stream / queue runs what
export TMPDIR=/tmp
O=gpurun_out/r06_21; mkdir -p $O
for cfg in "bair8:--batch 8" "bair64:"; do
  n=${cfg%%:*}; fl=${cfg#*:}
  rm -rf /tmp/tr_$n
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o t -- python bench.py --steps 4 --warmup 2 --lean $fl > $O/trace_$n.log 2>&1
  python tools/step_gaps.py /tmp/tr_$n 4 0 1.6 > $O/step_gaps_$n.txt 2>&1
  cp /tmp/tr_$n/t_kernel_trace.csv $O/kernel_trace_$n.csv 2>/dev/null || cp $(find /tmp/tr_$n -name "*kernel_trace.csv" | head -1) $O/kernel_trace_$n.csv
done
# the same loop without the profiler, pipelined vs serial, for reference
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --pipeline 0 2>/dev/null | tail -1 > $O/bench_bair8_serial.json
python -X faulthandler bench.py --batch 8 --steps 10 --warmup 2 --lean --emulate-collation rccl > $O/bench_bair8_rccl.json 2> $O/bench_bair8_rccl.err; echo "rccl-mode bench rc=$?" | tee -a $O/bench_bair8_rccl.err
tail -30 $O/bench_bair8_rccl.err
python - <<'PY'
import json
for f in ('bench_bair8','bench_bair8_serial'):
    r=json.load(open(f'gpurun_out/r06_21/{f}.json')); print(f, r['ms_per_step'], (r.get('single_call') or {}).get('ms'))
PY
sed -n 1,200p $O/step_gaps_bair8.txt | cut -c1-160
