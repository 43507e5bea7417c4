cd /root/repo; export TMPDIR=/tmp
for T in 208 204 4; do
echo "s01 TYB=$T: $(CICE_EVP_HIP_TYB=$T python bench.py --workload s01 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['us_per_subcycle'], d['roofline']['frac'])")"
done
for T in 204 4; do
echo "gx1 stream TYB=$T: $(CICE_EVP_HIP_RESIDENT=0 CICE_EVP_HIP_TYB=$T python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['us_per_subcycle'], d['roofline']['frac'])")"
done
