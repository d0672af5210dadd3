#!/bin/bash
# round 6, GPU call 31: statistics of the tiny maps without atomics / memset (one workgroup per sample and channel slice): A/B against the committed library + tests
export TMPDIR=/tmp
O=gpurun_out/r06_31; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_full.txt 2>&1; tail -3 $O/tests_full.txt
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export I2V_LIB_PATH=tools/_tl/libi2v_hip_prev.so; else unset I2V_LIB_PATH; fi
    timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64_$lib.csv 2>/dev/null | tail -1 > $O/bench_bair64_$lib.$rep.json
    timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_$lib.$rep.json
    timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_$lib.$rep.json
  done
done
unset I2V_LIB_PATH
rm -rf /tmp/tr_s; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_s -o t -- python bench.py --steps 3 --warmup 1 --lean > /dev/null 2>&1
grep -E "stats_kernel|coef_kernel|fillBuffer" $(find /tmp/tr_s -name "*kernel_stats.csv" | head -1) | cut -c1-160 > $O/kernel_stats_small.txt; cat $O/kernel_stats_small.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_31/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f,'ERR',e)
PY
