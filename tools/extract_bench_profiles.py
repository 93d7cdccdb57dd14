#!/usr/bin/env python3
"""Split a bench.py JSON line into the per-claim files under profiles/ (VERDICT r5 #1, #2, #6):
    <tag>_bench.json               the whole line
    <tag>_sharded_overhead.json    per-rank cost of the sharded step beside the unsharded one
    <tag>_controller_latency.json  BASELINE configs[0] through control.iLqr: control-step latency

    python tools/extract_bench_profiles.py gpurun_out/profiles_new/r06_bench.json r06 [outdir]"""
import json
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path(sys.argv[3]) if len(sys.argv) > 3 else src.parent
line = [l for l in src.read_text().splitlines() if l.startswith('{"metric"')][-1]
d = json.loads(line)
stamp = {"library_sha256": d["roofline"].get("library_sha256"), "source": f"{tag}_bench.json (python bench.py)"}
(out / f"{tag}_bench.json").write_text(line + "\n")
ex = d.get("extra", {})
if "sharded_overhead" in ex:
    (out / f"{tag}_sharded_overhead.json").write_text(json.dumps(
        {"_meta": stamp, "B1024": ex["sharded_overhead"], "B131072": ex.get("sharded_overhead_B131072")},
        indent=1) + "\n")
if "config1_closed_loop" in ex:
    (out / f"{tag}_controller_latency.json").write_text(json.dumps(
        {"_meta": stamp, **ex["config1_closed_loop"]}, indent=1) + "\n")
print("wrote", [p.name for p in out.glob(f"{tag}_*.json")])
