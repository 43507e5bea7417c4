cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r1r; mkdir -p $O
python bench.py > $O/bench_gx1.json 2> $O/bench_gx1.err; cut -c1-200 $O/bench_gx1.json
python bench.py --fused --no-cpu-baseline --no-secondary > $O/bench_gx1_fused.json 2>/dev/null
python bench.py --case caps --no-cpu-baseline --no-secondary > $O/bench_gx1_caps.json 2>/dev/null
python bench.py --workload gx3 --no-cpu-baseline --no-secondary > $O/bench_gx3.json 2>/dev/null
V=$(python -c "import json; print(json.load(open('$O/bench_gx1.json'))['config']['tile_variant'] % 1000)")
export CICE_EVP_HIP_RESIDENT=1 CICE_EVP_HIP_RES_GEN=2 CICE_EVP_HIP_RES_LOGW=$V
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gx1 -- python bench.py --no-cpu-baseline --no-secondary > $O/prof_gx1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
head -2 $O/prof/gx1_kernel_stats.csv | cut -c1-200; echo logw $V
