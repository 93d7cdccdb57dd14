#!/bin/bash
# One-shot refresh of profiles/ on the GPU box (run from the repo root through gpurun):
#   per workload:  rocprofv3 --kernel-trace --stats           -> kstats_<workload>.csv
#                  rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ cycle counters | SQ instruction counters |
#                  fp64 instruction counters (five SEPARATE passes, never combined with traces: MI355X_MICROARCH.md §HBM)
#   then           summarise (tools/summarise_profiles.py) -> pmc_traffic.json stamped with the
#                  sha256 of the library the counters were collected on
#                  bench.py JSON line + kernel stats of the bench command itself
# Results land in gpurun_out/profiles_new/ — copy them into profiles/ afterwards.
#   bash tools/collect_profiles.sh [round tag, default r04]
set -u
ROOT=$(pwd)
TAG=${1:-r06}
NEW=$ROOT/gpurun_out/profiles_new
RAW=$ROOT/gpurun_out/prof_raw
rm -rf "$NEW" "$RAW"; mkdir -p "$NEW" "$RAW"
cd /tmp && export TMPDIR=/tmp
# name            workload dtype batch    layout iters launches (>= 10 launches per kernel-trace row)
WORKLOADS="
config2_f64_B1024    config2 f64 1024    wave  10 24
config2_f64_B4096    config2 f64 4096    wave  10 24
config2_f64_B8192    config2 f64 8192    wave  10 24
config2_f64_B16384   config2 f64 16384   tiled 10 24
config2_f64_B32768   config2 f64 32768   tiled 10 24
config2_f64_B65536   config2 f64 65536   tiled 10 24
config2_f64_B131072  config2 f64 131072  tiled 10 24
config2_f32_B65536   config2 f32 65536   tiled 10 24
config2_f64_B1048576 config2 f64 1048576 tiled 10 12
config2_f32_B1048576 config2 f32 1048576 tiled 10 12
config5_f64_B65536   config5 f64 65536   tiled 4  16
config5_f32_B65536   config5 f32 65536   tiled 4  16
"
SQ_CYC="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
SQ_INS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
SQ_F64="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"
target() { echo python3 $ROOT/tools/pmc_target.py --workload $1 --dtype $2 --batch $3 --layout $4 --iters $5 --launches $6; }
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$RAW/calib_$ctr" -- python3 $ROOT/tools/pmc_calib.py \
    > "$RAW/calib_$ctr.log" 2>&1
done
echo "$WORKLOADS" | while read name wl dt b lay it ln; do
  [ -z "$name" ] && continue
  rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${name}_kstats" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_kstats.log" 2>&1
  find "$RAW/${name}_kstats" -name "*kernel_stats.csv" -exec cp {} "$NEW/${TAG}_kstats_${name}.csv" \;
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$RAW/${name}_FETCH_SIZE" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_FETCH_SIZE.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$RAW/${name}_WRITE_SIZE" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_WRITE_SIZE.log" 2>&1
  rocprofv3 --pmc $SQ_CYC --output-format csv -d "$RAW/${name}_SQCYC" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_SQCYC.log" 2>&1
  rocprofv3 --pmc $SQ_INS --output-format csv -d "$RAW/${name}_SQINS" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_SQINS.log" 2>&1
  rocprofv3 --pmc $SQ_F64 --output-format csv -d "$RAW/${name}_SQF64" -- $(target $wl $dt $b $lay $it $ln) \
    > "$RAW/${name}_SQF64.log" 2>&1
  echo "collected $name"
done
# solves to termination (kernel traces only): the speculative eight-lane kernel at 1024 problems,
# lane chunks + compaction + speculative tail at 65536
for spec in "1024 wave" "65536 tiled"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/solve_B$1_kstats" -- \
    python3 $ROOT/tools/pmc_target.py --workload config2 --dtype f64 --batch $1 --layout $2 --launches 10 --solve \
    > "$RAW/solve_B$1_kstats.log" 2>&1
  find "$RAW/solve_B$1_kstats" -name "*kernel_stats.csv" -exec cp {} "$NEW/${TAG}_kstats_solve_f64_B$1.csv" \;
done
cd "$ROOT"
python3 tools/summarise_profiles.py "$RAW" "$NEW" "$TAG" > "$NEW/summarise.log" 2>&1
cp "$NEW/pmc_traffic.json" "$ROOT/profiles/pmc_traffic.json"   # bench.py reads it from profiles/
python3 bench.py 2> "$NEW/bench.err" | grep '^{"metric"' | tail -1 > "$NEW/${TAG}_bench.json"
python3 tools/extract_bench_profiles.py "$NEW/${TAG}_bench.json" "$TAG" "$NEW"
# one solve of 65536 problems as a timeline of dispatches (tools/solve_timeline.py)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$RAW/solve_trace" -- python3 $ROOT/tools/pmc_target.py \
  --workload config2 --dtype f64 --batch 65536 --layout tiled --launches 5 --solve --sync-each > /dev/null 2>&1
cd "$ROOT"
python3 tools/solve_timeline.py "$RAW/solve_trace" --json "$NEW/${TAG}_solve_timeline.json" > "$NEW/${TAG}_solve_timeline.txt" 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/bench_kstats" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra > "$RAW/bench_kstats.log" 2>&1
find "$RAW/bench_kstats" -name "*kernel_stats.csv" -exec cp {} "$NEW/${TAG}_kstats_bench_headline.csv" \;
cd "$ROOT"
cat "$NEW/summarise.log" | tail -30
head -c 900 "$NEW/${TAG}_bench.json"; echo
