#!/bin/bash
# round 6, GPU call 3: the operand-GENERATING F(4,3) kernel (8 MFMA + 4 producer waves) on the two thin 128 x 128 shapes -- bits vs the
# F(4,3) kernel on the writer's operand, timing -- and the stream-configuration A/B with the collation stream emulated (4 vs 3 streams)
export TMPDIR=/tmp
O=gpurun_out/r06_3; mkdir -p $O
for s in "4 16 128 128 32 32 0 1 0 1" "4 16 128 128 64 32 0 0 0 2" "2 16 128 128 32 32 0 0 1 1" "32 16 128 128 32 32 0 1 0 1" "32 16 128 128 64 32 0 0 0 2"; do
  echo "== conv16w_check $s" >> $O/gen_check.txt
  timeout 600 tools/conv16w_check $s 2>&1 | grep -v "^$" >> $O/gen_check.txt
done
cat $O/gen_check.txt
for ss in own shared own shared; do
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair64_emu_$ss.$RANDOM.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair8_emu_$ss.$RANDOM.json
done
for ss in own shared; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --side-stream $ss 2>/dev/null | tail -1 > $O/bench_bair8_noemu_$ss.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_3/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f, 'ERR', e)
PY
