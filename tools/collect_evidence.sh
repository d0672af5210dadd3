# Re-creates the per-round evidence files (run on the GPU box: bash tools/collect_evidence.sh <outdir>); copy what is to be
# judged into profiles/ afterwards (named rNN_<letter>_...).
export TMPDIR=/tmp
out=${1:-gpurun_out/evidence}
mkdir -p $out
# the default line (BASELINE configs[1]) with every key, HBM traffic measured by the run itself (two rocprofv3 --pmc child passes)
I2V_PMC_OUT=$out/pmc_traffic timeout 900 python bench.py --live-traffic --per-layer $out/conv16_per_layer_bair64.csv 2>$out/bench_bair64.err | tail -1 > $out/bench_bair64.json
rm -rf $out/pmc_traffic/fetch_size $out/pmc_traffic/write_size
I2V_PMC_OUT=$out/pmc_traffic_land timeout 900 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --live-traffic --per-layer $out/conv16_per_layer_land128.csv 2>/dev/null | tail -1 > $out/bench_land128_b32.json
rm -rf $out/pmc_traffic_land/fetch_size $out/pmc_traffic_land/write_size
for b in 4 8 16; do timeout 200 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_bair64_b$b.json; done
timeout 300 python bench.py --config dtdb128 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_dtdb128_strong_b256.json
timeout 300 python bench.py --config iper128_t32 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_iper128_t32_strong_b128.json
timeout 300 python bench.py --config dtdb128 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_dtdb128_b32.json
timeout 300 python bench.py --config iper128_t32 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_iper128_t32_b16.json
# kernel traces of the timed steps only (--no-extras: no post-timing measurement loops in the trace)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bair -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/prof_bair.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_land -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/prof_land.log 2>&1
rm -f $out/prof_*/bench_kernel_trace.csv
# round 5 A/B legs (same box): in-call overlap off, cINN chain unfolded, exact-fp32 mode on the 27-tap kernel
I2V_DEC_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_bair64_overlap0.json
I2V_DEC_OVERLAP=0 timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_land128_overlap0.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact --small-batch 0 2>/dev/null | tail -1 > $out/bench_bair64_overlap1.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact 2>/dev/null | tail -1 > $out/bench_land128_overlap1.json
# round 6: the decoder in auto mode (per-layer fallback behind the range guard: a stream synchronisation per forward), and the stream
# configuration of an N > 1 rank emulated on this GPU (one shared side stream + the collation stream)
I2V_DEC_MMA=auto timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $out/bench_bair64_mma_auto.json
I2V_DEC_MMA=auto timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $out/bench_bair8_mma_auto.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $out/bench_bair8_lean.json
timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation 2>/dev/null | tail -1 > $out/bench_bair64_multi_gpu_streams.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation 2>/dev/null | tail -1 > $out/bench_bair8_multi_gpu_streams.json
I2V_DEC_WINO32=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --sustain 0 --small-batch 0 2>/dev/null | tail -1 > $out/bench_bair64_exact_direct.json
I2V_FLOW_FOLD=0 timeout 300 python tools/flowtime.py 2>&1 | grep -v amdgpu.ids > $out/flowtime_unfolded.txt
# cINN chain: latencies (fp32 and fp16-operand mode), per-kernel stats
timeout 300 python tools/flowtime.py 2>&1 | grep -v amdgpu.ids > $out/flowtime.txt
FLOWTIME_F16=1 timeout 300 python tools/flowtime.py 2>&1 | grep -v amdgpu.ids >> $out/flowtime.txt
FLOWTIME_B=64 timeout 300 bash tools/flow_prof.sh evidence_b64 > /dev/null 2>&1; cp gpurun_out/flowprof_evidence_b64.csv $out/kernel_stats_flow_b64.csv 2>/dev/null
# SQ counters of the F(4,3) kernel on the g_3.conv_1 shape (pass A and pass B are one kernel; MFMA-busy normalised by GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs)
timeout 400 bash tools/pmc_sq.sh $out/pmc_sq_f43 tools/conv16w_check 8 16 64 64 128 128 0 1 > $out/pmc_sq_f43_g3conv1.txt 2>&1
rm -rf $out/pmc_sq_f43
# per-workgroup phase timeline and per-tap timing of the F(4,3) kernel (instrumented builds: tools/build_measurement_libs.sh conv)
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 64 64 0 1" "4 16 128 128 32 32 0 1"; do
  [ -x tools/conv16w_check_tl ] && timeout 100 tools/conv16w_check_tl $s 2>&1 | grep -v "^$" >> $out/f43_timeline.txt
  [ -x tools/conv16w_check_tt ] && timeout 100 tools/conv16w_check_tt $s 2>&1 | grep -v "^$" >> $out/f43_taptime.txt
done
