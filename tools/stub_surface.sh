#!/bin/bash
# Which symbols of the Icepack interface stub (oracle/ref/icepack_intfc_stub.F90) do the reference's hot-path
# objects actually reference?  `nm` on the objects build_ref.sh compiled from the unmodified reference sources.
# Everything the EVP subcycle and its preparation take from Icepack must appear here, and nothing else can
# influence them: kinds and constants are compile-time (no symbol), the rest is this list.
set -euo pipefail
O="$(cd "$(dirname "$0")/.." && pwd)/oracle/_ref/obj_strict"
[ -d "$O" ] || { echo "oracle/_ref/obj_strict not built (oracle/ref/build_ref.sh strict)"; exit 0; }
for f in cap/ice_dyn_evp.o ice_dyn_shared.o ice_boundary.o ice_grid.o ice_dyn_core1d.o; do
  echo "== $f"
  nm -u "$O/$f" 2>/dev/null | grep -i icepack | sed 's/^ *U //' | sort -u || true
done
