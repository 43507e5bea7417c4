cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1g
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "degenerate or s01_full" > gpurun_out/r1g/pytest_edge.log 2>&1 ) 2>&1 | grep real; grep -E "passed|failed|^E " gpurun_out/r1g/pytest_edge.log | tail -15
