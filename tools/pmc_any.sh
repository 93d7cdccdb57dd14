#!/bin/bash
# Arbitrary counter groups (one --pmc pass per quoted group) on one kernel configuration:
#   bash tools/pmc_any.sh config5 65536 f64 wave 4 "SQ_WAIT_ANY SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum"
set -u
ROOT=$(pwd)
WL=$1; B=$2; DT=$3; LAY=$4; IT=$5; shift 5
OUT=$ROOT/gpurun_out/pmcany
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 $ROOT/tools/pmc_target.py \
    --workload $WL --layout $LAY --batch $B --dtype $DT --launches 2 --iters $IT > "$OUT/g$i.log" 2>&1 || tail -3 "$OUT/g$i.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "iterate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v) / len(v):16.0f}  (per launch, {len(v)} launches)")
PY
