cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1j
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r1j/pytest_gpu.log 2>&1; grep -E "passed|failed|^E |Error|s call" gpurun_out/r1j/pytest_gpu.log | tail -12
