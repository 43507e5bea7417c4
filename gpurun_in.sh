cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|differ|Error" | tail -8
for v in "CICE_EVP_HIP_SELF_EXCHANGE=1" "CICE_EVP_HIP_SELF_EXCHANGE=1 CICE_EVP_HIP_OVERLAP=1" "CICE_EVP_HIP_SELF_EXCHANGE=1 CICE_EVP_HIP_GRAPH_RCCL=1" "CICE_EVP_HIP_SELF_EXCHANGE=1 CICE_EVP_HIP_GRAPH_RCCL=1 CICE_EVP_HIP_OVERLAP=1"; do echo "$v"; env $v timeout 120 python gpurun_selfx.py 2>&1 | grep -E "RESULT|rror" | cut -c1-80; done
