#!/bin/bash
# HBM traffic of the dominant kernels from PMC counters, collected exactly as
# MI355X_MICROARCH.md §HBM / §rocprofv3 prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
# passes (TCC has 4 slots: FETCH_SIZE needs 3, WRITE_SIZE 2), never combined with traces, plus a
# calibration pass on a streaming copy in the same access widths.  Run on the GPU box from the
# repo root; writes gpurun_out/pmc/*.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name counter -- target args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${name}_${ctr}" -- python3 "$@" \
    > "$OUT/${name}_${ctr}.log" 2>&1
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run calib $ctr $ROOT/tools/pmc_calib.py
  run wave_f64_B1024 $ctr $ROOT/tools/pmc_target.py --layout wave --batch 1024 --dtype f64
  run lane_f64_B65536 $ctr $ROOT/tools/pmc_target.py --layout lane --batch 65536 --dtype f64
  run tiled_f32_B65536 $ctr $ROOT/tools/pmc_target.py --layout tiled --batch 65536 --dtype f32
  run tiled_f64_B1048576 $ctr $ROOT/tools/pmc_target.py --layout tiled --batch 1048576 --dtype f64 --launches 2
done
# third pass: where the wavefronts' cycles go (SQ counters, quad-cycles; MI355X_MICROARCH.md PMC table:
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES) — still no trace flags beside --pmc
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
runsq() {
  local name=$1; shift
  rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${name}_SQ" -- python3 "$@" \
    > "$OUT/${name}_SQ.log" 2>&1
}
runsq wave_f64_B1024 $ROOT/tools/pmc_target.py --layout wave --batch 1024 --dtype f64
runsq lane_f64_B65536 $ROOT/tools/pmc_target.py --layout lane --batch 65536 --dtype f64
runsq tiled_f32_B65536 $ROOT/tools/pmc_target.py --layout tiled --batch 65536 --dtype f32
runsq tiled_f64_B1048576 $ROOT/tools/pmc_target.py --layout tiled --batch 1048576 --dtype f64 --launches 2
python3 $ROOT/tools/summarise_pmc.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
