#!/usr/bin/env python3
"""The per-workload table of profiles/README.md from profiles/pmc_traffic.json (kernel-trace
averages, algorithmic fraction of the 8 TB/s peak, PMC bytes over algorithmic bytes):
    python tools/profiles_table.py [profiles/pmc_traffic.json]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
d = json.loads(Path(sys.argv[1] if len(sys.argv) > 1 else ROOT / "profiles" / "pmc_traffic.json").read_text())
ALG = {("config2", "f64"): 4968, ("config2", "f32"): 2484, ("config5", "f64"): 33912,
       ("config5", "f32"): 16956}
print("library", d["_meta"]["lib_sha256"][:16])
print("| workload | kernel | kernel-trace avg (calls) | M it/s | frac of 8 TB/s (algorithmic) | PMC bytes ÷ algorithmic | SQ: issuing / parked / stalled |")
print("|---|---|---|---|---|---|---|")
rows = []
for key, r in d.items():
    if key.startswith(("_", "solve")):
        continue
    wl, dt, B, it = key.split(":")
    B, it = int(B[1:]), int(it[2:])
    t = r["kernel_avg_ms_kernel_trace"] * 1e-3
    alg = ALG[(wl, dt)]
    sq = r.get("sq_shares_of_wave_cycles") or {}
    rows.append((wl, dt != "f64", B, f"| {wl} {dt} B={B} | `{r['kernel']}` | "
                 f"{r['kernel_avg_ms_kernel_trace'] * 1e3:.1f} µs ({r['kernel_calls_kernel_trace']}) | "
                 f"{B * it / t / 1e6:.0f} | {alg * B * it / t / 8e12:.3f} | "
                 f"{r.get('hbm_bytes_per_problem_iteration', 0) / alg:.2f} | "
                 f"{sq.get('issuing_any_instruction', 0):.2f} / {sq.get('parked_on_waitcnt_or_barrier', 0):.2f} / "
                 f"{sq.get('issue_stalled', 0):.2f} |"))
for row in sorted(rows):
    print(row[3])
for key in ("solve:f64:B1024", "solve:f64:B65536"):
    if key in d:
        print(key, f"{d[key]['kernels_ms_per_solve']:.3f} ms of kernels per solve;",
              {k: round(v, 3) for k, v in d[key]["kernel_time_shares"].items()})
