cd /root/repo; export TMPDIR=/tmp
export CICE_EVP_HIP_RESIDENT=1 CICE_EVP_HIP_RES_GEN=2 CICE_EVP_HIP_RES_LOGW=4
for D in 0 4 1 2; do
echo "gx1 dbg=$D: $(CICE_EVP_HIP_RES_DEBUG=$D python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['us_per_subcycle'])")"
done
for D in 0 1 2; do
echo "gx3 dbg=$D: $(CICE_EVP_HIP_RES_DEBUG=$D python bench.py --workload gx3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['us_per_subcycle'])")"
done
