cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1j
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "between_processes or prep" > gpurun_out/r1j/pytest_prep2.log 2>&1; grep -E "passed|failed|^E |Error" gpurun_out/r1j/pytest_prep2.log | tail -15
