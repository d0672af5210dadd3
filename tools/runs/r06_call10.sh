#!/bin/bash
# round 6, GPU call 10: the LOADER form of the thin F(4,3) layers (four extra waves issue the V requests) vs the round-5 kernels
export TMPDIR=/tmp
O=gpurun_out/r06_10; mkdir -p $O
for s in "8 16 128 128 32 32 0 1" "8 16 128 128 64 32 0 0" "32 16 128 128 32 32 0 1" "32 16 128 128 64 32 0 0"; do
  for l in 1 0 1 0; do
    echo "== loader=$l $s: $(I2V_W4_LOADER=$l timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'F\(4,3\)' | tr -s ' ' | tr '\n' '|')" >> $O/loader_ab.txt
  done
  echo "== loader=0 nth=512 $s: $(I2V_W4_LOADER=0 I2V_W4_NTH=512 timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'F\(4,3\) +[0-9]' | tr -s ' ')" >> $O/loader_ab.txt
done
cat $O/loader_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "structure_switches or full_width_128 or nf8_128 or model_128 or cfg5 or determinism_soak or generated_operand" 2>&1 | tail -5 > $O/gpu_tests_subset.txt; cat $O/gpu_tests_subset.txt
for k in 1 2 3; do
  timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --per-layer $O/per_layer_land128.csv 2>/dev/null | tail -1 > $O/bench_land128_loader.$k.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_10/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'), 'frac', r['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
grep g_4 $O/per_layer_land128.csv
