cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1n
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rehearsal" > gpurun_out/r1n/pytest_reh.log 2>&1; grep -E "passed|failed|^E |Error" gpurun_out/r1n/pytest_reh.log | tail -12
