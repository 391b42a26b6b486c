mkdir -p gpurun_out
L=artdeco_amd/lib
(
for rep in 1 2; do
 for v in libartdeco_hip.so libartdeco_hip.ssimhw.so; do
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 ssim_fwd,ssim_bwd 2>&1 | tail -1
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 512 384 ssim_fwd,ssim_bwd 2>&1 | tail -1
 done
done
) > gpurun_out/r05_ab_ssim_xcd.txt 2>&1
(timeout 2400 python -m pytest tests -m gpu -q -s -x > gpurun_out/r05_gputests_a.log 2>&1; echo "rc $?" >> gpurun_out/r05_gputests_a.log)
cat gpurun_out/r05_ab_ssim_xcd.txt
grep -E "step-oracle.*(knife rows|settle|oracle [0-9])|passed|failed|^E  |^FAILED|rc " gpurun_out/r05_gputests_a.log | cut -c1-700 | tail -40
