export CICE_EVP_HIP_VERBOSE=1
for p in "450 2400" "900 1200"; do
  for m in 1 0; do
    CICE_EVP_HIP_TESTING=1 CICE_EVP_HIP_MARCH_SELFX=1 CICE_EVP_HIP_MARCH_DIRECT=$m timeout 300 python tools/piece_timing.py $p 96 5 --comm 2>&1 | grep "PIECE\|ring\|trial\|fault\|rror" | sed 's/\[.*\]//'
  done
  timeout 300 python tools/piece_timing.py $p 96 5 2>&1 | grep "PIECE" | sed 's/\[.*\]//'
done
