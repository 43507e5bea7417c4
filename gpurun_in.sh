cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|differ|Error" | tail -5
P='import sys,json; r=json.loads(sys.stdin.read()); print(round(r["value"]/1e9,3), round(r["config"]["us_per_subcycle"],2), round(r["roofline"]["frac"],3), r["config"].get("tile_variant"))'
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"; done
python bench.py --no-cpu-baseline --fused 2>/dev/null | python -c "$P"
python bench.py --no-cpu-baseline --workload s01 --steps 2 --warmup 1 2>/dev/null | python -c "$P"
