#!/bin/bash
# Measurement builds (run in the build container; the .so / binaries travel to the GPU box with gpurun):
#   tools/_tl/libi2v_hip_nopreload.so the library without -amdgpu-kernarg-preload-count
#   tools/_tl/libi2v_hip_flowtl.so   the library with -DFLOW_TIMELINE (per-launch / per-phase stamps of the cINN tile chain,
#                                    read by tools/flow_timeline.py)
#   tools/conv16w_check[_tl|_tt]     the conv check tool: plain, -DW4_TIMELINE, -DW4_TAPTIME (all with -DI2V_MEASURE: the I2V_W4_* switches)
# The build that reads the F(4,3) structure switches (I2V_W4_PIPE / SKEW / ORDER / BN / NTH / TRACE, I2V_CONVIMG_TCH) is
#   image2video-synthesis-using-cinns_amd/lib/libi2v_hip_measure.so   (make -C .../csrc measure; also built by __graft_entry__.build())
# -- the production library reads no environment variable on a launch path.
set -e
cd "$(dirname "$0")/.."
CS=image2video-synthesis-using-cinns_amd/csrc
mkdir -p tools/_tl
make -C $CS -j4 measure
if [ "$1" != "conv" ]; then
  ALL="$CS/i2v_common.hip $CS/i2v_flow.hip $CS/i2v_flow_tile.hip $CS/i2v_conv.hip $CS/i2v_wino32.hip $CS/i2v_pointwise.hip $CS/i2v_conv16.hip $CS/i2v_conv16w.hip $CS/i2v_conv16w4.hip $CS/i2v_conv16w4g.hip $CS/i2v_convimg.hip $CS/i2v_dec.hip $CS/i2v_embed.hip $CS/i2v_encoder.hip $CS/i2v_ops.hip"
  # the shipped library WITHOUT kernarg preload (A/B of that flag: FLOWTIME_LIB=tools/_tl/libi2v_hip_nopreload.so python tools/flowtime.py)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -shared -I$CS -Iinclude $ALL -o tools/_tl/libi2v_hip_nopreload.so &
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFLOW_TIMELINE -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -shared -I$CS -Iinclude \
    $CS/i2v_common.hip $CS/i2v_flow.hip $CS/i2v_flow_tile.hip $CS/i2v_conv.hip $CS/i2v_wino32.hip $CS/i2v_pointwise.hip $CS/i2v_conv16.hip $CS/i2v_conv16w.hip \
    $CS/i2v_conv16w4.hip $CS/i2v_conv16w4g.hip $CS/i2v_convimg.hip $CS/i2v_dec.hip $CS/i2v_embed.hip $CS/i2v_encoder.hip $CS/i2v_ops.hip -o tools/_tl/libi2v_hip_flowtl.so &
fi
if [ "$1" = "nt" ]; then
  # cache-policy experiments: V stream non-temporal / output stores non-temporal / operand writer non-temporal
  ALL="$CS/i2v_common.hip $CS/i2v_flow.hip $CS/i2v_flow_tile.hip $CS/i2v_conv.hip $CS/i2v_wino32.hip $CS/i2v_pointwise.hip $CS/i2v_conv16.hip $CS/i2v_conv16w.hip $CS/i2v_conv16w4.hip $CS/i2v_conv16w4g.hip $CS/i2v_convimg.hip $CS/i2v_dec.hip $CS/i2v_embed.hip $CS/i2v_encoder.hip $CS/i2v_ops.hip"
  for v in "vnt:-DW4_V_NT" "outnt:-DW4_OUT_NT" "modnt:-DMOD_NT" "allnt:-DW4_V_NT -DW4_OUT_NT -DMOD_NT"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${v#*:} -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -shared -I$CS -Iinclude $ALL -o tools/_tl/libi2v_hip_${v%%:*}.so &
  done
  wait; ls -la tools/_tl; exit 0
fi
if [ "$1" != "flow" ]; then
  # tools/conv16w_check*: the check tool linked against the library's own objects (csrc/build: `make` + `make measure`), so that every
  # translation unit keeps its flags (i2v_conv16w4g.o: -fno-slp-vectorize); only the F(4,3) unit is recompiled for the instrumented variants
  make -C $CS -j4
  B=$CS/build
  HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DI2V_MEASURE -I$CS -Iinclude"
  $HC -c tools/conv16w_check.hip -o tools/_tl/conv16w_check.o &
  $HC -DW4_TIMELINE -c tools/conv16w_check.hip -o tools/_tl/conv16w_check_tl.o &
  $HC -DW4_TAPTIME -c tools/conv16w_check.hip -o tools/_tl/conv16w_check_tt.o &
  $HC -DW4_TIMELINE -c $CS/i2v_conv16w4.hip -o tools/_tl/i2v_conv16w4_tl.o &
  $HC -DW4_TAPTIME -c $CS/i2v_conv16w4.hip -o tools/_tl/i2v_conv16w4_tt.o &
  # ablations of the operand-generating kernel's producer role (results wrong, timing only): 1 nothing generated, 2 no global loads,
  # 3 no LDS stores, 4 loads only
  for v in 1 2 3 4; do $HC -fno-slp-vectorize -DW4G_ABLATE=$v -c $CS/i2v_conv16w4g.hip -o tools/_tl/i2v_conv16w4g_abl$v.o & done
  wait
  REST="$B/i2v_conv16w.o $B/i2v_conv16.o $B/i2v_common.o"
  L="/opt/rocm/bin/hipcc --offload-arch=gfx950"
  $L tools/_tl/conv16w_check.o $B/i2v_conv16w4.m.o $B/i2v_conv16w4g.o $REST -o tools/conv16w_check &
  $L tools/_tl/conv16w_check_tl.o tools/_tl/i2v_conv16w4_tl.o $B/i2v_conv16w4g.o $REST -o tools/conv16w_check_tl &
  $L tools/_tl/conv16w_check_tt.o tools/_tl/i2v_conv16w4_tt.o $B/i2v_conv16w4g.o $REST -o tools/conv16w_check_tt &
  for v in 1 2 3 4; do $L tools/_tl/conv16w_check.o $B/i2v_conv16w4.m.o tools/_tl/i2v_conv16w4g_abl$v.o $REST -o tools/conv16w_check_abg$v & done
fi
wait
ls -la tools/_tl tools/conv16w_check*
