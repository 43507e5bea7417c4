cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|differ|Error|rror" | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|rror"
