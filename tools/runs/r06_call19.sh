#!/bin/bash
# round 6, GPU call 19: kernel trace of the pipelined loop -> idle gaps of the main queue inside one step (B = 64, B = 8, Landscape)
export TMPDIR=/tmp
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_19; mkdir -p $O
for cfg in "bair64:" "bair8:--batch 8" "land128:--config land128"; do
  n=${cfg%%:*}; fl=${cfg#*:}
  rm -rf /tmp/tr_$n
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o t -- python bench.py --steps 4 --warmup 2 --lean $fl > $O/trace_$n.log 2>&1
  python tools/step_gaps.py /tmp/tr_$n 4 > $O/step_gaps_$n.txt 2>&1
  head -60 $O/step_gaps_$n.txt
done
