#!/bin/bash
# Round 6: the three forms of the marching path's ring exchange ranked on ONE GPU (the ring exchanged with the rank itself across
# the cyclic seam: CICE_EVP_HIP_MARCH_SELFX=1 -- pack, transfer, unpack and the launch structure are those of a real run; the
# transfer is a device copy), on the 8x1 and 4x2 pieces of 3600x2400, and the piece's subcycles per pass (2 / 3 / 4).
#   after the pass (product default) | overlapped with the rest of the pass (OVERLAP=1) | direct stores, no RCCL (DIRECT=1) | none
cd "$(dirname "$0")/.."
for p in "450 2400" "900 1200"; do
  for form in "" "CICE_EVP_HIP_MARCH_OVERLAP=1" "CICE_EVP_HIP_MARCH_DIRECT=1"; do
    env CICE_EVP_HIP_MARCH_SELFX=1 $form timeout 300 python tools/piece_timing.py $p 96 5 --comm 2>&1 | grep "PIECE" | sed "s/^PIECE/RING [${form:-after the pass (default)}]/"
  done
  timeout 300 python tools/piece_timing.py $p 96 5 2>&1 | grep "PIECE" | sed 's/^PIECE/RING [no exchange]/'
  for k in 2 3 4; do
    env CICE_EVP_HIP_MARCH_K=$k timeout 300 python tools/piece_timing.py $p 96 5 2>&1 | grep "PIECE" | sed "s/^PIECE/KPASS $k [no exchange]/"
  done
done
