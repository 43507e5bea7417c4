cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1q
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r1q/pytest_gpu.log 2>&1; grep -E "passed|failed|^E  .*rror" gpurun_out/r1q/pytest_gpu.log | tail -5
python tools/soak.py gx1 5000 2>&1 | grep SOAK
