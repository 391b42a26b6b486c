mkdir -p gpurun_out
L=artdeco_amd/lib
(timeout 1500 python -m pytest tests/test_step_oracle.py -m gpu -q -s > gpurun_out/r05_step_oracle4.log 2>&1; echo "rc $?" >> gpurun_out/r05_step_oracle4.log)
(timeout 300 python tools/lab/split_gemm_lab.py > gpurun_out/r05_split_gemm_lab.txt 2>&1)
(
for rep in 1 2; do
 for v in libartdeco_hip.so libartdeco_hip.ssimhw.so; do
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 ssim_fwd,ssim_bwd,raster_bwd 2>&1 | tail -1
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 512 384 ssim_fwd,ssim_bwd,raster_bwd 2>&1 | tail -1
 done
done
) > gpurun_out/r05_ab_ssim_xcd.txt 2>&1
(timeout 600 python -m pytest tests/test_native_step.py tests/test_ssim.py tests/test_fused_glue.py -m gpu -q > gpurun_out/r05_misc_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05_misc_tests.log)
grep -E "step-oracle.*(knife rows|settle|oracle [0-9]|phases)|passed|failed|^E  " gpurun_out/r05_step_oracle4.log | cut -c1-900
cat gpurun_out/r05_split_gemm_lab.txt | tail -16; cat gpurun_out/r05_ab_ssim_xcd.txt; tail -4 gpurun_out/r05_misc_tests.log
