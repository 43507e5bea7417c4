cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1l
( time python bench.py > gpurun_out/r1l/bench_gx1.json 2> gpurun_out/r1l/bench_gx1.err ) 2>&1 | grep real; python -c "
import json; d=json.load(open('gpurun_out/r1l/bench_gx1.json')); print(d['value'], d['config']['us_per_subcycle'], d['roofline']['frac']); print(d.get('secondary')); print(d.get('tripole')); print(d['cpu_baseline']['value'])"
