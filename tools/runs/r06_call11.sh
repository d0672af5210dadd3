#!/bin/bash
# round 6, GPU call 11: the round's evidence set on ONE box (tools/collect_evidence.sh), and where the generating kernel's outputs differ
export TMPDIR=/tmp
bash tools/collect_evidence.sh gpurun_out/r06_n > gpurun_out/r06_n.log 2>&1
for s in "4 16 128 128 32 32 0 1 0 1"; do echo "== $s"; timeout 300 tools/conv16w_check $s 2>&1 | grep -B6 -E 'GEN \(mode'; done > gpurun_out/r06_n/gen_diff_hist.txt 2>&1
tail -5 gpurun_out/r06_n.log; ls gpurun_out/r06_n | head -60
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_n/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'), 'frac', (r.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
