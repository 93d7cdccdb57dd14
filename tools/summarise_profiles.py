#!/usr/bin/env python3
"""Reduce the raw rocprofv3 outputs of tools/collect_profiles.sh.

  pmc_traffic.json           per workload key (bench.py's "<workload>:<dtype>:B<batch>:it<iters>"):
                             HBM bytes per launch (FETCH_SIZE x calibration + WRITE_SIZE), shares of
                             the wavefronts' lifetime (SQ cycle counters), wavefront-instructions per
                             launch by type (SQ instruction counters) and the kernel's average
                             duration from the --kernel-trace --stats pass; "_meta" carries the
                             sha256 of the library the counters were collected on
  <tag>_pmc_summary.json     the same plus the raw counter values and the calibration factors

Units and corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; the read
counter is calibrated on streaming copies of known size in the same access width (8 B or 4 B per
lane) and the measured ratio (2.0 on gfx950) is applied; WRITE_SIZE is exact."""
import csv
import glob
import hashlib
import json
import sys
from collections import defaultdict
from pathlib import Path

raw, new, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "ilqr_iterative_tasks_amd" / "csrc" / "libi2lqr_hip.so"


def counters(pattern):
    res = defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            res[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return res


def avg(v):
    return sum(v) / len(v)


calib = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for (kname, c), v in counters(f"{raw}/calib_{ctr}/**/*counter_collection.csv").items():
        if "k_copy" in kname and c == ctr:
            elem = 8 if "double" in kname else 4
            calib[(ctr, elem)] = ((1 << 28) * elem) / (avg(v) * 1024.0)

out = {"_meta": {"lib_sha256": hashlib.sha256(LIB.read_bytes()).hexdigest(),
                 "source": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ cycle "
                           "counters / SQ instruction counters in four separate passes per workload; "
                           "FETCH_SIZE x calibration factor measured on streaming copies of known size "
                           "(2.0 on gfx950), WRITE_SIZE exact; all values per launch",
                 "calibration_known_over_reported":
                     {f"{k[0]}_{k[1]}B_per_lane": v for k, v in calib.items()}}}
full = {"_meta": out["_meta"]}
for d in sorted(glob.glob(f"{raw}/*_kstats")):
    name = Path(d).name[: -len("_kstats")]
    if name.startswith("solve"):
        # solves to termination (kernel traces only): where the time of one solve goes, by kernel
        B = int(name.split("_B")[1])
        fam = defaultdict(float)
        launches = 10  # collect_profiles.sh: --launches 10
        for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "i2lqr::" in r["Name"]:
                    fam[r["Name"].split("<")[0].replace("void ", "").replace("i2lqr::", "")] += \
                        float(r["TotalDurationNs"])
        tot = sum(fam.values())
        if tot:
            rec = {"kernels_ms_per_solve": tot / launches / 1e6,
                   "kernel_time_shares": {k: v / tot for k, v in sorted(fam.items())},
                   "outputs": "U, X, lamb, cost, iters, status (no gains: what the reference's "
                              "ilqr() returns)"}
            out[f"solve:f64:B{B}"] = rec
            full[f"solve:f64:B{B}"] = rec
            print(f"solve:f64:B{B}", json.dumps(rec))
        continue
    if name == "bench":
        continue  # kernel trace only
    wl, dt, bb = name.split("_")
    B = int(bb[1:])
    iters = 4 if wl == "config5" else 10
    elem = 8 if dt == "f64" else 4
    key = f"{wl}:{dt}:B{B}:it{iters}"
    rec, rawrec = {}, {}
    # the dominant kernel of the trace = the iterate kernel of this workload
    kname, kavg, kcalls = None, None, None
    for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "iterate" in r["Name"]:
                kname, kavg, kcalls = r["Name"], float(r["AverageNs"]), int(r["Calls"])
                kmin, kmax = float(r.get("MinNs", 0) or 0), float(r.get("MaxNs", 0) or 0)
                break
    if kname:
        rec["kernel"] = kname.split("<")[0].replace("void ", "").replace("i2lqr::", "")
        rec["kernel_avg_ms_kernel_trace"] = kavg / 1e6
        rec["kernel_calls_kernel_trace"] = kcalls
        rec["kernel_min_ms_kernel_trace"], rec["kernel_max_ms_kernel_trace"] = kmin / 1e6, kmax / 1e6
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for (kn, c), v in counters(f"{raw}/{name}_{ctr}/**/*counter_collection.csv").items():
            if "iterate" in kn and c == ctr:
                rawrec[ctr + "_KiB"] = avg(v)
                rec[ctr.lower() + "_bytes"] = avg(v) * 1024.0 * calib.get((ctr, elem), 2.0 if ctr == "FETCH_SIZE" else 1.0)
    if "fetch_size_bytes" in rec and "write_size_bytes" in rec:
        rec["hbm_bytes_per_launch"] = rec["fetch_size_bytes"] + rec["write_size_bytes"]
        rec["hbm_bytes_per_problem_iteration"] = rec["hbm_bytes_per_launch"] / (B * iters)
    sq = {}
    for part in ("SQCYC", "SQINS"):
        for (kn, c), v in counters(f"{raw}/{name}_{part}/**/*counter_collection.csv").items():
            if "iterate" in kn:
                sq[c] = avg(v)
    # fifth pass: fp64 arithmetic instructions by kind -> EXECUTED flops per launch (64 lanes per
    # wavefront-instruction, a multiply-add counted as two; an upper bound where lanes are masked)
    f64 = {}
    for (kn, c), v in counters(f"{raw}/{name}_SQF64/**/*counter_collection.csv").items():
        if "iterate" in kn:
            f64[c] = avg(v)
    if f64:
        rawrec["sq_f64"] = f64
        rec["executed_fp64_flops_per_launch"] = 64.0 * (
            2.0 * f64.get("SQ_INSTS_VALU_FMA_F64", 0) + f64.get("SQ_INSTS_VALU_ADD_F64", 0) +
            f64.get("SQ_INSTS_VALU_MUL_F64", 0) + f64.get("SQ_INSTS_VALU_TRANS_F64", 0))
        rec["executed_fp64_flops_per_problem_iteration"] = rec["executed_fp64_flops_per_launch"] / (B * iters)
    rawrec["sq"] = sq
    if sq.get("SQ_WAVE_CYCLES"):
        wc = sq["SQ_WAVE_CYCLES"]
        rec["sq_shares_of_wave_cycles"] = {
            "issuing_any_instruction": sq.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "issuing_valu": sq.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "parked_on_waitcnt_or_barrier": sq.get("SQ_WAIT_ANY", 0) / wc,
            "issue_stalled": sq.get("SQ_WAIT_INST_ANY", 0) / wc}
    ins = {k: v for k, v in sq.items() if k.startswith("SQ_INSTS_") and "F64" not in k}
    if ins:
        rec["wave_instructions_by_type"] = ins
        rec["wave_instructions_per_launch"] = sum(ins.values())
    if sq.get("SQ_WAVES"):
        rec["waves_per_launch"] = sq["SQ_WAVES"]
    out[key] = rec
    full[key] = dict(rec, raw=rawrec)
    print(key, json.dumps({k: v for k, v in rec.items() if not isinstance(v, dict)}))
json.dump(out, open(f"{new}/pmc_traffic.json", "w"), indent=1)
json.dump(full, open(f"{new}/{tag}_pmc_summary.json", "w"), indent=1)
