#!/bin/bash
# round 6, GPU call 4: mma = auto tests, the shared-stream order with an explicit prepare, the default line with both small-batch configurations
export TMPDIR=/tmp
O=gpurun_out/r06_4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "mma_auto or range_guard or model_forward_semantics or from_pixels or shared_side or generate_samples or both_matrix" 2>&1 | tail -12 > $O/gpu_tests_subset.txt
cat $O/gpu_tests_subset.txt
for ss in own shared; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair8_emu_$ss.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair64_emu_$ss.json
done
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --side-stream shared 2>/dev/null | tail -1 > $O/bench_bair8_noemu_shared.json
I2V_DEC_MMA=auto timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_mma_auto.json
I2V_DEC_MMA=auto timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_mma_auto.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_default.json
( time timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json ) 2> $O/bench_default.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_4/bench_*.json')):
    try:
        r=json.load(open(f)); sb=r.get('small_batch') or {}
        print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'), r.get('streams',{}).get('side_stream'))
        if sb: print('   small_batch n1', sb.get('n1'), '\n   multi', sb.get('multi_gpu'), '\n   proj', {k:v for k,v in sb['projected_strong_scaling'].items() if k in ('single_call','single_call_before_collation','pipelined','all_gather_ms_model')})
        if 'config_128' in r: print('   config_128', {k:v for k,v in r['config_128'].items() if k in ('ms_per_step','frames_per_s','seconds','error')})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/bench_default.time; tail -3 $O/bench_default.err
