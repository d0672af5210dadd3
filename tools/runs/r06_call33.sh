#!/bin/bash
# round 6, GPU call 33: the GPU suite and the driver's bench command with the final library (loader MODE 3 instantiated, production paths unchanged)
export TMPDIR=/tmp
O=gpurun_out/r06_33; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_full.txt 2>&1; tail -3 $O/tests_full.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json ) 2> $O/bench_driver_cmd.time
python - <<'PY'
import json
r=json.load(open('gpurun_out/r06_33/bench_driver_cmd.json'))
print('ms/step', r['ms_per_step'], 'value', r['value'], 'roofline', r['roofline']['frac'], 'cfg128', r['config_128']['ms_per_step'])
PY
tail -3 $O/bench_driver_cmd.time
