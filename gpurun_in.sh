cd /root/repo; export TMPDIR=/tmp
python tools/run_call_overhead.py 2>&1 | grep -E "RESULT|rror"
timeout 900 python -m pytest tests -x -q -m gpu -k "dropin or golden_strict" 2>&1 | grep -E "passed|failed|differ|Error|rror" | tail -4
