#!/bin/bash
# round 6, GPU call 2: measurement-switch worker test, shared-side-stream test, A/B of the stream configurations (own side stream of the
# decoder handle next to the prefetch stream vs ONE shared side stream), and the default line with its config_128 child leg
export TMPDIR=/tmp
O=gpurun_out/r06_2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "structure_switches or shared_side or capture or lifetime or overlap_keeps or prepare or pipelined" 2>&1 | tail -8 > $O/gpu_tests_subset.txt
cat $O/gpu_tests_subset.txt
for ss in own shared own shared; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact --no-config-128 --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair64_$ss.$RANDOM.json
done
for ss in own shared; do
  I2V_PREFETCH_PRIO=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact --no-config-128 --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair64_${ss}_prio0.json
  timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --side-stream $ss 2>/dev/null | tail -1 > $O/bench_land128_$ss.json
done
( time timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json ) 2> $O/bench_default.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_2/bench_*.json')):
    try:
        r=json.load(open(f)); sb=r.get('small_batch') or {}
        print(f.split('/')[-1], 'ms/step %.2f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'), 'B8 single', sb.get('single_call_ms'), 'B8 piped', sb.get('pipelined_ms_per_step'), 'proj', (sb.get('projected_strong_scaling') or {}).get('single_call'), (sb.get('projected_strong_scaling') or {}).get('pipelined'))
        if 'config_128' in r: print('   config_128', {k:v for k,v in r['config_128'].items() if k in ('ms_per_step','frames_per_s','seconds','error')})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/bench_default.time
