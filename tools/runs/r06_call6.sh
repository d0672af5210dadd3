#!/bin/bash
# round 6, GPU call 6: where does the generating kernel's time go?  Ablations of the producer role (results wrong, timing only), the MFMA
# role's stand-alone 512-thread kernel for reference, the decoder-level bit test and the Landscape step with I2V_DEC_GEN=0 / 1
export TMPDIR=/tmp
O=gpurun_out/r06_6; mkdir -p $O
for s in "8 16 128 128 32 32 0 1 0 1" "8 16 128 128 64 32 0 0 0 2"; do
  for b in conv16w_check conv16w_check_abg1 conv16w_check_abg2 conv16w_check_abg3 conv16w_check_abg4; do
    echo "== $b $s: $(timeout 300 tools/$b $s 2>&1 | grep -E ' GEN  ' | tr -s ' ')" >> $O/gen_ablation.txt
  done
  echo "== F(4,3) 256-thread $s: $(timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'F\(4,3\) +[0-9]' | tr -s ' ')" >> $O/gen_ablation.txt
  echo "== F(4,3) 512-thread $s: $(I2V_W4_NTH=512 timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'F\(4,3\) +[0-9]' | tr -s ' ')" >> $O/gen_ablation.txt
done
cat $O/gen_ablation.txt
timeout 600 python -m pytest tests -m gpu -x -q -s -k "generated_operand" 2>&1 | tail -8 > $O/gpu_test_gen.txt
cat $O/gpu_test_gen.txt
for g in 0 1 0 1; do
  I2V_DEC_GEN=$g timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --per-layer $O/per_layer_land128_gen$g.csv 2>/dev/null | tail -1 > $O/bench_land128_gen$g.$RANDOM.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_6/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f, 'ERR', e)
PY
grep g_4 $O/per_layer_land128_gen*.csv
