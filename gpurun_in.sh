cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1s
export CICE_EVP_HIP_RES_DEBUG=8
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tripole_seam_bitwise or tx1_size or resident_kernel_golden" > gpurun_out/r1s/lagtest.log 2>&1; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r1s/lagtest.log | cut -c1-250 | head -20
