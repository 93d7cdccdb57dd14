#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_iters; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for it in 1 2 4 10; do for ctr in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $ctr --output-format csv -d $OUT/it${it}_$ctr -- python3 $ROOT/tools/pmc_target.py --layout lane --batch 65536 --dtype f64 --iters $it --launches 3 > $OUT/it${it}_$ctr.log 2>&1
done; done
python3 - <<PY
import csv,glob
for it in (1,2,4,10):
    tot=0
    for ctr,f in (("FETCH_SIZE",2.0),("WRITE_SIZE",1.0)):
        v=[float(r["Counter_Value"]) for fn in glob.glob("$OUT/it%d_%s/**/*counter_collection.csv"%(it,ctr),recursive=True) for r in csv.DictReader(open(fn)) if "iterate" in r["Kernel_Name"]]
        b=sum(v)/len(v)*1024*f; tot+=b
        print(it,ctr,"%.1f B/problem"%(b/65536))
    print(it,"total %.1f B/problem"%(tot/65536))
PY
