cd /root/repo; export TMPDIR=/tmp
python tools/prep_timing.py gx1 2>&1 | grep -E "PREP|EVPCALL"
