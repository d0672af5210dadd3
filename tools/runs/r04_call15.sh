#!/bin/bash
# round 4, GPU call 15: sample tiles per hidden-layer workgroup of the cINN tile chain at B = 64 / 128
export TMPDIR=/tmp
for ns in 1 2 4; do
  FLOWTIME_B=64,128 I2V_FLOW_NS=$ns timeout 200 python tools/flowtime.py 2>&1 | grep -v amdgpu.ids | awk '{print "NS", $18, "B", $8, $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
done
