cd /root/repo; export TMPDIR=/tmp
P='import sys,json; r=json.loads(sys.stdin.read()); print(round(r["value"]/1e9,3), round(r["config"]["us_per_subcycle"],2), round(r["roofline"]["frac"],3), r["config"].get("tile_variant"))'
for v in "CICE_EVP_HIP_RESIDENT=0" "CICE_EVP_HIP_RESIDENT=1" "A=1"; do echo -n "gx1 $v: "; env $v python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"; done
echo -n "gx1 fused resident: "; CICE_EVP_HIP_RESIDENT=1 python bench.py --no-cpu-baseline --fused 2>/dev/null | python -c "$P"
echo -n "gx1 caps auto: "; python bench.py --no-cpu-baseline --case caps 2>/dev/null | python -c "$P"
echo -n "gx3 auto: "; python bench.py --no-cpu-baseline --workload gx3 2>/dev/null | python -c "$P"
