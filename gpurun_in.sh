cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r1/pytest_gpu.log; tail -2 gpurun_out/r1/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r1/bench_gx1.json 2> gpurun_out/r1/bench_gx1.err; cat gpurun_out/r1/bench_gx1.json
python bench.py --fused --no-cpu-baseline > gpurun_out/r1/bench_gx1_fused.json 2>/dev/null
python bench.py --workload s01 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/bench_s01.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1/prof -o gx1 -- python bench.py --no-cpu-baseline > gpurun_out/r1/prof_gx1.log 2>&1
CICE_EVP_HIP_NOGRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1/pmc -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_fetch.log 2>&1
CICE_EVP_HIP_NOGRAPH=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1/pmc -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_write.log 2>&1
ls gpurun_out/r1/prof gpurun_out/r1/pmc
