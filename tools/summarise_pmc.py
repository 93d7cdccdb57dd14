#!/usr/bin/env python3
"""Reduce the FETCH_SIZE / WRITE_SIZE passes of tools/collect_pmc.sh to per-launch HBM bytes.

Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; the read counter is
calibrated on a streaming copy of known size in the same access width (8 B or 4 B per lane) and
the measured ratio is applied as the correction factor (the guide's x2 holds for 16 B/lane)."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = sys.argv[1]


def per_kernel(pattern):
    res = defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            res[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return res


def avg(v):
    return sum(v) / len(v)


summary = {}
calib = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    res = per_kernel(f"{out}/calib_{ctr}/**/*counter_collection.csv")
    for (kname, c), v in res.items():
        if "k_copy" in kname:
            elem = 8 if "double" in kname else 4
            known = (1 << 28) * elem
            calib[(ctr, elem)] = known / (avg(v) * 1024.0)
summary["calibration_known_over_reported"] = {f"{k[0]}_{k[1]}B_per_lane": v for k, v in calib.items()}
for name, elem in (("wave_f64_B1024", 8), ("lane_f64_B65536", 8), ("tiled_f32_B65536", 4),
                   ("tiled_f64_B1048576", 8)):
    rec = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        res = per_kernel(f"{out}/{name}_{ctr}/**/*counter_collection.csv")
        for (kname, c), v in res.items():
            if "iterate" in kname:
                rec[ctr + "_KiB_raw"] = avg(v)
                rec[ctr + "_bytes_corrected"] = avg(v) * 1024.0 * calib.get((ctr, elem), 1.0)
    if rec:
        rec["hbm_bytes_per_launch"] = rec.get("FETCH_SIZE_bytes_corrected", 0) + \
            rec.get("WRITE_SIZE_bytes_corrected", 0)
    # SQ pass: shares of the wavefronts' lifetime (all values per launch, summed over the chip)
    sq = {}
    for (kname, c), v in per_kernel(f"{out}/{name}_SQ/**/*counter_collection.csv").items():
        if "iterate" in kname:
            sq[c] = avg(v)
    if sq.get("SQ_WAVE_CYCLES"):
        wc = sq["SQ_WAVE_CYCLES"]
        rec["sq"] = {k: v for k, v in sq.items()}
        rec["sq_shares_of_wave_cycles"] = {
            "issuing_any_instruction": sq.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "issuing_valu": sq.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "parked_on_waitcnt_or_barrier": sq.get("SQ_WAIT_ANY", 0) / wc,
            "issue_stalled": sq.get("SQ_WAIT_INST_ANY", 0) / wc}
        if sq.get("SQ_INSTS_VALU") is not None:
            tot = sq["SQ_INSTS_VALU"] + sq.get("SQ_INSTS_SALU", 0)
            rec["valu_fraction_of_valu_plus_salu_instructions"] = sq["SQ_INSTS_VALU"] / max(tot, 1)
    if rec:
        summary[name] = rec
print(json.dumps(summary, indent=1))
