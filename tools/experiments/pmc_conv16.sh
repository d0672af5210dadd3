# (uses ablation builds of the direct-weights variant: hipcc -DI2V_ABLATE=n ... -o tools/conv16_bench_a<n>; see README.md)
# development helper: clock / MFMA-busy of the ablation builds of tools/conv16_bench (see I2V_ABLATE in i2v_conv16.hip)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc7
for m in 0 1 2 7; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc7/a$m -o pmc -- ./tools/conv16_bench_a$m 8 > gpurun_out/pmc7/a$m.log 2>&1 || echo "pass $m failed"
done
timeout 120 rocprofv3 --pmc VmemLatency LdsLatency MemUnitStalled MfmaUtil TA_BUSY_avr --kernel-trace --output-format csv -d gpurun_out/pmc7/lat -o pmc -- ./tools/conv16_bench_a0 8 > gpurun_out/pmc7/lat.log 2>&1 || echo "pass lat failed"
