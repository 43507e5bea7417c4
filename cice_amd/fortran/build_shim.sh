#!/bin/bash
# Compile check of the Fortran host side (ISO_C_BINDING shim + Option-B forwarding
# module) with amdflang.  The shim `use`s CICE modules (ice_kinds_mod, ice_blocks,
# ice_domain, ice_grid, ice_dyn_shared, ...): their .mod files come from the in-place
# reference build under oracle/_ref/obj_strict (oracle/ref/build_ref.sh).  Without
# them (GPU box, fresh clone) there is nothing to check against and the step is skipped.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
MODS="$REPO/oracle/_ref/obj_strict"
FC=${FC:-/opt/rocm/bin/amdflang}
if [ ! -f "$MODS/ice_dyn_shared.mod" ] || [ ! -x "$FC" ]; then
  echo "build_shim: CICE module files not available -- skipped" >&2
  exit 0
fi
OUT="$REPO/oracle/_ref/shimcheck"
mkdir -p "$OUT"
cd "$OUT"
$FC -O2 -cpp -I"$MODS" -c "$HERE/ice_dyn_evp_hip.F90" -o ice_dyn_evp_hip.o 2>/dev/null
$FC -O2 -cpp -I"$MODS" -I. -c "$HERE/ice_dyn_evp1d_hip.F90" -o ice_dyn_evp1d.o 2>/dev/null
echo "build_shim: ice_dyn_evp_hip.F90, ice_dyn_evp1d_hip.F90 compile OK"
