cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1g
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "remote or mailbox or self_exchange" > gpurun_out/r1g/pytest_remote.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r1g/pytest_remote.log | tail -15
