#!/bin/bash
# round 6, GPU call 9: where do the generating kernel's outputs differ from the writer path; the direct kernel's temporal tap skipping (g_0.conv_1)
export TMPDIR=/tmp
O=gpurun_out/r06_9; mkdir -p $O
for s in "4 16 128 128 32 32 0 1 0 1" "4 16 128 128 64 32 0 0 0 2"; do echo "== $s"; timeout 300 tools/conv16w_check $s 2>&1 | grep -A5 -E 'GEN \(mode'; done > $O/gen_diff_hist.txt 2>&1
cat $O/gen_diff_hist.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "full_width or nf8 or shape_sweep or baseline_batch or b8_every_row or cfg1 or tile_width or alternative_kernel or mma_auto or negative_sigma" 2>&1 | tail -5 > $O/gpu_tests_subset.txt; cat $O/gpu_tests_subset.txt
timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --per-layer $O/per_layer_bair8.csv 2>/dev/null | tail -1 > $O/bench_bair8.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --per-layer $O/per_layer_land128.csv 2>/dev/null | tail -1 > $O/bench_land128.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_9/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f, 'ERR', e)
PY
head -5 $O/per_layer_bair64.csv; head -5 $O/per_layer_bair8.csv; head -5 $O/per_layer_land128.csv
