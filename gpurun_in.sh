cd /root/repo; export TMPDIR=/tmp
python tools/soak.py gx1 20000 2>&1 | grep SOAK
python tools/soak.py tx1 20000 2>&1 | grep SOAK
python tools/soak.py gx3 20000 2>&1 | grep SOAK
