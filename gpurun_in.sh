cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r1n
export CICE_EVP_HIP_HALO_TIMEOUT_MS=60000
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 tools/mailbox_2proc.py --workload gx3 --ndte 24 --shape 2x4 --expect-resident --timing > gpurun_out/r1n/mb8.log 2>&1 ) 2>&1 | grep real
grep MAILBOX gpurun_out/r1n/mb8.log | cut -c1-700
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29712 tools/mailbox_2proc.py --workload gx1 --ndte 24 --shape 2x4 --expect-resident --timing > gpurun_out/r1n/mb8b.log 2>&1 ) 2>&1 | grep real
grep MAILBOX gpurun_out/r1n/mb8b.log | cut -c1-700
