cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r1f/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r1f/pytest_gpu.log | tail -5
export CICE_EVP_HIP_SELF_EXCHANGE=1 CICE_EVP_HIP_HALO=direct CICE_EVP_HIP_HALO_RIDE=0 CICE_EVP_HIP_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1f/prof_mb2 -o mb -- python tools/selfx_timing.py s01 > gpurun_out/r1f/prof_mb2.log 2>&1
head -4 gpurun_out/r1f/prof_mb2/mb_kernel_stats.csv | cut -c1-160
