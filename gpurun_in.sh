cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1k
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r1k/pytest_gpu.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r1k/pytest_gpu.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-160
