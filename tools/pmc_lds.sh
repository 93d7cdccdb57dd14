#!/bin/bash
# LDS / issue counters of one kernel configuration (one --pmc pass per counter group):
#   bash tools/pmc_lds.sh config5 65536 f64 wave 4
set -u
ROOT=$(pwd)
WL=${1:-config5}; B=${2:-65536}; DT=${3:-f64}; LAY=${4:-wave}; IT=${5:-4}
OUT=$ROOT/gpurun_out/pmclds
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 $ROOT/tools/pmc_target.py \
    --workload $WL --layout $LAY --batch $B --dtype $DT --launches 2 --iters $IT > "$OUT/g$i.log" 2>&1
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
