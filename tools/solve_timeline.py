#!/usr/bin/env python3
"""Timeline of ONE solve to termination from a rocprofv3 kernel trace: every dispatch of the last
solve of the traced run with its start offset, duration and the gap to the previous dispatch's end.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- \
        python3 tools/pmc_target.py --workload config2 --dtype f64 --batch 65536 --layout tiled \
        --launches 6 --solve
    python3 tools/solve_timeline.py <dir> [--json out.json]

The solves are separated by their first kernel (the fill of the live counters / the first chunk):
a new solve starts where a dispatch follows a gap of more than 200 us."""
import csv
import json
import re
import sys
from pathlib import Path

root = Path(sys.argv[1])
files = sorted(root.rglob("*kernel_trace.csv"))
assert files, f"no kernel_trace.csv under {root}"
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    s = m.group(1) if m else name[:40]
    if "k_group_spec" in name or "k_lane_iterate" in name or "k_group_iterate" in name:
        t = re.search(r"<(.*)>", name)
        if t:
            args = [a.strip() for a in t.group(1).split(",")]
            s += "<" + ",".join(a for a in args[2:]) + ">"
    return s


solves, cur = [], []
for i, (s, e, n) in enumerate(rows):
    if cur and s - cur[-1][1] > 200_000:
        solves.append(cur)
        cur = []
    cur.append((s, e, n))
if cur:
    solves.append(cur)
last = solves[-1]
t0 = last[0][0]
out = []
prev_end = t0
print(f"{len(solves)} groups of dispatches; the last one: {len(last)} dispatches, "
      f"{(last[-1][1] - t0) / 1e3:.1f} us from first start to last end")
busy = 0
for s, e, n in last:
    rec = {"kernel": short(n), "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3,
           "gap_us": (s - prev_end) / 1e3}
    out.append(rec)
    busy += e - s
    print(f"{rec['start_us']:9.1f}  {rec['dur_us']:8.1f}  gap {rec['gap_us']:6.1f}  {rec['kernel']}")
    prev_end = max(prev_end, e)
print(f"kernels busy {busy / 1e3:.1f} us, gaps {(last[-1][1] - t0 - busy) / 1e3:.1f} us")
if "--json" in sys.argv:
    Path(sys.argv[sys.argv.index("--json") + 1]).write_text(json.dumps(
        {"dispatches": out, "total_us": (last[-1][1] - t0) / 1e3, "busy_us": busy / 1e3}, indent=1))
