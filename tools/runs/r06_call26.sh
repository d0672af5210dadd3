#!/bin/bash
# round 6, GPU call 26: the collation of an N > 1 rank through a ONE-rank RCCL group (torch.distributed issues the gather) on the prefetch stream vs the copy
# stand-in vs no collation; the driver's command line
export TMPDIR=/tmp
O=gpurun_out/r06_26; mkdir -p $O
J() { grep '^{' | tail -1; }
for k in 1 2; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | J > $O/bench_bair8_one_gpu.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation copy 2>/dev/null | J > $O/bench_bair8_copy_on_prefetch.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl 2>/dev/null | J > $O/bench_bair8_rccl1_on_prefetch.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl --collation-stream own 2>/dev/null | J > $O/bench_bair8_rccl1_own_side_shared.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl --collation-stream own --side-stream own 2>/dev/null | J > $O/bench_bair8_rccl1_own_side_own.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | J > $O/bench_bair64_one_gpu.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation rccl 2>/dev/null | J > $O/bench_bair64_rccl1_on_prefetch.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation rccl --collation-stream own --side-stream own 2>/dev/null | J > $O/bench_bair64_rccl1_own_side_own.$k.json
done
timeout 300 python bench.py --batch 8 --steps 10 --warmup 2 --lean --emulate-collation rccl 2>/dev/null | tail -3 | cut -c1-120 > $O/stdout_tail_rccl_mode.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json ) 2> $O/bench_driver_cmd.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_26/bench_*.json')):
    try:
        r=json.load(open(f)); st=r.get('streams',{})
        print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'), st.get('side_stream'), '|', st.get('collation_stream'), '|', st.get('collation_stream_emulated'))
    except Exception as e: print(f,'ERR',e)
PY
cat $O/stdout_tail_rccl_mode.txt; cat $O/bench_driver_cmd.time
