#!/bin/bash
# round 6, GPU call 12: diagnosis of the generating kernel's last-bit differences (which stage: modulation, transform / split, layout), graph replay probe
export TMPDIR=/tmp
O=gpurun_out/r06_12; mkdir -p $O
for m in 0 1 2 3; do
  echo "== I2V_CHECK_GEN=$m: $(I2V_CHECK_GEN=$m timeout 300 tools/conv16w_check 2 16 128 128 32 32 0 0 1 1 2>&1 | grep -E 'GEN \(mode' | tr -s ' ')" >> $O/gen_diag.txt
done
cat $O/gen_diag.txt
timeout 600 python tools/graph_replay_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/graph_replay_probe.txt
