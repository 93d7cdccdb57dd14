"""GPU suite: bench.py honours the driver's contract — one JSON line with the BASELINE metric, the
roofline object of the dominant kernel and the CPU baseline — and __graft_entry__.smoke() passes."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-extra", "--cpu-seconds", "1"], capture_output=True, text=True,
                         timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("batched iLQR iterations/s") and d["unit"] == "iLQR iterations/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] or "config2" in d["config"]["workload"]
    assert d["config"]["batch_per_gpu"] == 1024 and d["config"]["iterations_per_step"] == 10
    assert d["value"] > 1e6  # north_star floor: 1e6 batched iLQR iterations/s on one MI355X
    assert abs(d["value"] - 1024 * 10 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_bytes_per_iteration"] == 4968
    assert abs(r["achieved"] - 4968 * 1024 * 10 / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout
