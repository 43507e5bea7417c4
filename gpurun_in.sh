cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tripole or trip" > gpurun_out/r1f/pytest_trip.log 2>&1; grep -E "passed|failed" gpurun_out/r1f/pytest_trip.log | tail -5
