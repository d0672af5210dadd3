#!/bin/bash
# round 6, GPU call 28: what the driver runs at round end, on the final commit: GPU suite, smoke(), the bench command line
export TMPDIR=/tmp
O=gpurun_out/r06_28; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_full.txt 2>&1; tail -3 $O/tests_full.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver_cmd.err | tail -1 > $O/bench_driver_cmd.json ) 2> $O/bench_driver_cmd.time
python - <<'PY'
import json
r=json.load(open('gpurun_out/r06_28/bench_driver_cmd.json'))
print('ms/step', r['ms_per_step'], 'value', r['value'], 'roofline', r['roofline']['frac'], 'cpu', r['cpu_baseline']['value'], 'cfg128', r['config_128']['ms_per_step'], 'proj', r['small_batch']['projected_strong_scaling']['single_call'], r['small_batch']['projected_strong_scaling']['pipelined'])
PY
cat $O/bench_driver_cmd.time
