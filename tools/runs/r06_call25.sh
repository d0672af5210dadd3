#!/bin/bash
# round 6, GPU call 25: full GPU suite with the direct-conv changes; cINN prefetch stream priority A/B through bench.py
export TMPDIR=/tmp
O=gpurun_out/r06_25; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_full.txt 2>&1; tail -4 $O/tests_full.txt
for rep in 1 2; do
  for pr in -1 0; do
    I2V_PREFETCH_PRIO=$pr timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_prio$pr.$rep.json
    I2V_PREFETCH_PRIO=$pr timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_prio$pr.$rep.json
    I2V_PREFETCH_PRIO=$pr timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_prio$pr.$rep.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_25/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f,'ERR',e)
PY
