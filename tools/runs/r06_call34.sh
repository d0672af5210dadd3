#!/bin/bash
# round 6, GPU call 34: priority of the decoder handle's side stream (SPADE branches, shortcuts): lowest / default / highest
export TMPDIR=/tmp
O=gpurun_out/r06_34; mkdir -p $O
for rep in 1 2; do
  for pr in 1 0 -1; do
    I2V_DEC_SIDE_PRIO=$pr timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_sideprio$pr.$rep.json
    I2V_DEC_SIDE_PRIO=$pr timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_sideprio$pr.$rep.json
    I2V_DEC_SIDE_PRIO=$pr timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_sideprio$pr.$rep.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_34/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'), 'dominant frac %.3f'%r['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
