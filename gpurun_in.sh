cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|differ|Error|rror|^tests.*Error|def test_" | tail -30
