cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1j
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r1j/pytest_dropin.log 2>&1; grep -E "passed|failed|^E |Error" gpurun_out/r1j/pytest_dropin.log | tail -15
