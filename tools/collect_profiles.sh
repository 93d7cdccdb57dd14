#!/bin/bash
# One-shot refresh of profiles/ on the GPU box (run from the repo root through gpurun):
#   1. HBM traffic of the dominant kernels (tools/collect_pmc.sh: separate --pmc passes)
#   2. the bench.py JSON line (reads the fresh traffic numbers)
#   3. rocprofv3 --kernel-trace --stats of the same bench command (no counters in this pass)
# Results land in gpurun_out/profiles_new/ — copy them into profiles/ afterwards.
set -u
ROOT=$(pwd)
NEW=$ROOT/gpurun_out/profiles_new
mkdir -p "$NEW"
bash tools/collect_pmc.sh > "$NEW/collect_pmc.log" 2>&1
cp "$ROOT/gpurun_out/pmc/summary.json" "$NEW/pmc_hbm_traffic_summary.json"
python3 - "$NEW/pmc_hbm_traffic_summary.json" "$ROOT/profiles/pmc_traffic.json" <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
out = {"_source": "tools/collect_pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
       "FETCH_SIZE x2.0 calibrated on 8 B/lane and 4 B/lane streaming copies of known size; "
       "WRITE_SIZE exact); bytes per launch of 10 fused iterations"}
for key, src, kern in (("config2:f64:B1024:it10", "wave_f64_B1024", "k_iterate (wave)"),
                       ("config2:f64:B65536:it10", "lane_f64_B65536",
                        "k_lane_iterate (batch-minor; tiled measures the same)"),
                       ("config2:f32:B65536:it10", "tiled_f32_B65536", "k_lane_iterate (tiled)"),
                       ("config2:f64:B1048576:it10", "tiled_f64_B1048576", "k_lane_iterate (tiled)")):
    out[key] = {"hbm_bytes_per_launch": s[src]["hbm_bytes_per_launch"], "kernel": kern}
    if "sq_shares_of_wave_cycles" in s[src]:  # third --pmc pass: SQ counters
        out[key]["sq_shares_of_wave_cycles"] = s[src]["sq_shares_of_wave_cycles"]
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
cp "$ROOT/profiles/pmc_traffic.json" "$NEW/pmc_traffic.json"
python3 bench.py 2> "$NEW/bench.err" | tail -1 > "$NEW/bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$NEW/kstats" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --steps 20 > "$NEW/kstats.log" 2>&1
find "$NEW/kstats" -name "*kernel_stats.csv" -exec cp {} "$NEW/kernel_stats_bench.csv" \;
rm -rf "$NEW/kstats"
head -c 600 "$NEW/bench.json"; echo; head -8 "$NEW/kernel_stats_bench.csv"
