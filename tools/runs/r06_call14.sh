#!/bin/bash
# round 6, GPU call 14: shared side stream with the first two SPADE levels inline and the cINN pass of step k + 1 in front of decoder k's side work
export TMPDIR=/tmp
O=gpurun_out/r06_14; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "shared_side or overlap_keeps or prepare or pipelined or capture or lifetime" 2>&1 | tail -4
for k in 1 2; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation --side-stream shared 2>/dev/null | tail -1 > $O/bench_bair8_emu_shared.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation --side-stream shared 2>/dev/null | tail -1 > $O/bench_bair64_emu_shared.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_own.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_own.$k.json
done
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --side-stream shared 2>/dev/null | tail -1 > $O/bench_bair8_noemu_shared.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --emulate-collation --side-stream shared 2>/dev/null | tail -1 > $O/bench_land128_emu_shared.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_14/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f, 'ERR', e)
PY
