#!/bin/bash
# Quick HBM-traffic check of one lane-kernel configuration (FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes, as tools/collect_pmc.sh does):
#   bash tools/pmc_quick.sh tiled 1048576 f64 [launches] [option=value ...]
set -u
ROOT=$(pwd)
LAY=${1:-tiled}; B=${2:-1048576}; DT=${3:-f64}; L=${4:-2}; shift 4 || true
OUT=$ROOT/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$ctr" -- python3 $ROOT/tools/pmc_target.py \
    --layout $LAY --batch $B --dtype $DT --launches $L --options "$*" > "$OUT/$ctr.log" 2>&1
done
python3 - "$OUT" $B "$LAY $DT $*" <<'PY'
import csv, glob, sys
out, B, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "iterate" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals.append(float(r["Counter_Value"]))
    tot[ctr] = sum(vals) / max(len(vals), 1) * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
per = {k: v / (B * 10) for k, v in tot.items()}
print(f"B={B} {tag}: read {per['FETCH_SIZE']:.0f} B  write {per['WRITE_SIZE']:.0f} B  total "
      f"{per['FETCH_SIZE'] + per['WRITE_SIZE']:.0f} B per problem-iteration "
      f"(algorithmic 4968 fp64 / 2484 fp32)")
PY
