for p in "450 2400" "900 1200" "900 2400" "1800 1200"; do
  for s in 0 10 14 20 28 40 60 100; do
    CICE_EVP_HIP_MARCH_SEG=$s timeout 120 python tools/piece_timing.py $p 96 5 2>&1 | grep PIECE
  done
done
