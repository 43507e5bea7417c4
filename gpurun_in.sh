cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1d
export CICE_EVP_HIP_RESIDENT=1 CICE_EVP_HIP_RES_LOGW=5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1d/prof -o gx1 -- python bench.py --no-cpu-baseline > gpurun_out/r1d/prof_gx1.log 2>&1
grep -h value gpurun_out/r1d/prof_gx1.log | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1d/pmc -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1d/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1d/pmc -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1d/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r1d/pmc -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1d/pmc_sq.log 2>&1
head -3 gpurun_out/r1d/prof/gx1_kernel_stats.csv
