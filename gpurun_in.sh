cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1g
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r1g/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r1g/pytest_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
