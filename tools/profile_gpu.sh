#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace statistics and PMC passes of the three
# kernel/workload pairs the bench line's roofline block quotes, written under gpurun_out/<tag>/.
# tools/pmc_summary.py reduces them to the tracked summary profiles/<round>_pmc_summary.json.
#
# Autotuning probes are switched off (forced kernel choice) so that every launch in a trace is a real one.
# Counters are collected in their own passes, with --kernel-trace only (no --stats, no other
# trace domain), as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes: SQ 8 slots per pass,
# FETCH_SIZE and WRITE_SIZE cannot share a pass, GRBM 2 slots.
#   usage: tools/profile_gpu.sh <tag> [what ...]     what: gx1res gx1str s01str s01march cgx1 cgs01 cgx1one cgx1res cgtx1res cgtx1 calib (default: all but cg*)
#   cgx1 / cgs01: the three kernels of the C-grid subcycle (tools/cgrid_timing.py) -- trace, HBM-byte and SQ passes;
#   cgx1one: the one-launch kernel (cg_one, the default on gx1) -- cgx1 pins the three-launch form
#   cgtx1res / cgtx1: the tripole grid tx1 with the resident kernel's FOLD variant / as five phases + fold steps
set -u
TAG=${1:-prof}; shift || true
WHAT=${*:-gx1res gx1str s01str calib}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"

SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"

run_passes () {   # name, env assignments (string), command...
  local name=$1 envs=$2; shift 2
  # 1. kernel trace + stats (durations; no counters in this pass)
  env $envs rocprofv3 --kernel-trace --stats -d "$OUT" -o "${name}_trace" --output-format csv -- "$@" > "$OUT/${name}_trace.log" 2>&1
  # 2. counters, one block family per pass
  env $envs rocprofv3 --pmc $SQ1 --kernel-trace -d "$OUT" -o "${name}_sq1" --output-format csv -- "$@" > "$OUT/${name}_sq1.log" 2>&1
  env $envs rocprofv3 --pmc $SQ2 --kernel-trace -d "$OUT" -o "${name}_sq2" --output-format csv -- "$@" > "$OUT/${name}_sq2.log" 2>&1
  env $envs rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT" -o "${name}_fetch" --output-format csv -- "$@" > "$OUT/${name}_fetch.log" 2>&1
  env $envs rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --kernel-trace -d "$OUT" -o "${name}_write" --output-format csv -- "$@" > "$OUT/${name}_write.log" 2>&1
  grep -h '^{"metric"' "$OUT/${name}_trace.log" | tail -1 > "$OUT/${name}_bench_under_trace.json"
}

for w in $WHAT; do
  case $w in
    gx1res) run_passes gx1res "CICE_EVP_HIP_RESIDENT=1 CICE_EVP_HIP_RES_LOGW=${RES_LOGW:-4} CICE_EVP_HIP_TYB=4" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary ;;
    gx1str) run_passes gx1str "CICE_EVP_HIP_RESIDENT=0 CICE_EVP_HIP_TYB=${TYB_GX1:-4}" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary ;;
    s01march) run_passes s01march "CICE_EVP_HIP_MARCH=1 ${MARCH_ENV:-}" python bench.py --workload s01 --ndte 96 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ;;
    s01str) run_passes s01str "CICE_EVP_HIP_MARCH=0 CICE_EVP_HIP_RESIDENT=0 CICE_EVP_HIP_TYB=${TYB_S01:-208}" python bench.py --workload s01 --ndte 24 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary ;;
    cgx1|cgs01|cgx1one|cgs01one|cgx1res|cgtx1res|cgtx1)
            g=gx1; [ $w = cgs01 ] || [ $w = cgs01one ] && g=s01
            [ $w = cgtx1res ] || [ $w = cgtx1 ] && g=tx1
            nd=120; [ $g = s01 ] && nd=12
            export CICE_EVP_HIP_CGRID_ONE=0; [ $w = cgx1one ] || [ $w = cgs01one ] || [ $w = cgx1res ] && export CICE_EVP_HIP_CGRID_ONE=1
            # cgx1res: the on-chip resident kernel (cg_res: one launch per call); the others pin the per-subcycle kernels
            export CICE_EVP_HIP_CGRID_RESIDENT=0; [ $w = cgx1res ] || [ $w = cgtx1res ] && export CICE_EVP_HIP_CGRID_RESIDENT=1
            rocprofv3 --kernel-trace --stats -d "$OUT" -o "${w}_trace" --output-format csv -- python tools/cgrid_timing.py $g --ndte $nd --reps 2 > "$OUT/${w}_trace.log" 2>&1
            rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT" -o "${w}_fetch" --output-format csv -- python tools/cgrid_timing.py $g --ndte $nd --reps 1 > "$OUT/${w}_fetch.log" 2>&1
            rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --kernel-trace -d "$OUT" -o "${w}_write" --output-format csv -- python tools/cgrid_timing.py $g --ndte $nd --reps 1 > "$OUT/${w}_write.log" 2>&1
            rocprofv3 --pmc $SQ1 --kernel-trace -d "$OUT" -o "${w}_sq1" --output-format csv -- python tools/cgrid_timing.py $g --ndte $nd --reps 1 > "$OUT/${w}_sq1.log" 2>&1
            rocprofv3 --pmc $SQ2 --kernel-trace -d "$OUT" -o "${w}_sq2" --output-format csv -- python tools/cgrid_timing.py $g --ndte $nd --reps 1 > "$OUT/${w}_sq2.log" 2>&1 ;;
    calib)  # known byte counts in the streaming kernels' access width (8 B per lane): calibrates FETCH_SIZE / WRITE_SIZE
            hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib || continue
            /tmp/pmc_calib > "$OUT/calib_plain.log" 2>&1
            for c in FETCH_SIZE WRITE_SIZE; do
              rocprofv3 --pmc $c --kernel-trace -d "$OUT" -o "calib_$c" --output-format csv -- /tmp/pmc_calib > "$OUT/calib_$c.log" 2>&1
            done ;;
  esac
done
ls "$OUT" | head -100
