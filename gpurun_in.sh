cd /root/repo; export TMPDIR=/tmp
python tools/hbm_triad.py 2>&1 | grep HBM
