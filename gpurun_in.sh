cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1b
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|differ|Error" | tail -4 | tee gpurun_out/r1b/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py > gpurun_out/r1b/bench_gx1.json 2> gpurun_out/r1b/bench_gx1.err; cat gpurun_out/r1b/bench_gx1.json | cut -c1-600
python bench.py --fused --no-cpu-baseline > gpurun_out/r1b/bench_gx1_fused.json 2>/dev/null
python bench.py --workload s01 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/bench_s01.json 2>/dev/null
python bench.py --case caps --no-cpu-baseline > gpurun_out/r1b/bench_gx1_caps.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1b/prof -o gx1 -- python bench.py --no-cpu-baseline > gpurun_out/r1b/prof_gx1.log 2>&1
V=$(python -c "import json; print(json.load(open('gpurun_out/r1b/bench_gx1.json'))['config']['tile_variant'])")
CICE_EVP_HIP_TYB=$V CICE_EVP_HIP_NOGRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1b/pmc -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/pmc_fetch.log 2>&1
CICE_EVP_HIP_TYB=$V CICE_EVP_HIP_NOGRAPH=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1b/pmc -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/pmc_write.log 2>&1
CICE_EVP_HIP_TYB=$V CICE_EVP_HIP_NOGRAPH=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r1b/pmc -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/pmc_sq.log 2>&1
echo variant $V
