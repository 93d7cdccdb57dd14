"""CPU suite: properties of the generated gfx950 code that cost round 3 a tenth of the quad12
kernel's throughput before they were noticed in the ISA: generic-address-space
(flat_*) memory operations — 64-bit vector addresses, counted in both vmcnt and lgkmcnt — and
scratch (spill) traffic inside the horizon loops.  Compiles two translation units to ISA with hipcc
(cross-compiles without a GPU; ~80 s side by side)."""
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parent.parent / "ilqr_iterative_tasks_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
         "--cuda-device-only"]


def _isa(tu: str, out_dir: Path) -> str:
    out = out_dir / (tu + ".s")
    subprocess.run([HIPCC, *FLAGS, "-o", str(out), str(CSRC / (tu + ".hip"))], check=True,
                   capture_output=True, timeout=900)
    return out.read_text()


def _loops(text: str, kernel_prefix: str):
    """{loop header: [instructions]} of the first kernel whose symbol starts with kernel_prefix
    (blocks attributed to the loop the compiler's annotation names, as tools/isa_loops.py does)."""
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel_prefix) and ":" in l)
    loops, cur = {}, None
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", l)
        if m:
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", m.group(2))
            cur = ".L" + h.group(1) if h else (m.group(1) if "Loop Header" in m.group(2) or "Parent Loop" in m.group(2) else None)
            continue
        if cur and l.startswith("\t") and not l.startswith(("\t;", "\t.")):
            loops.setdefault(cur, []).append(l.split()[0])
    return loops


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("isa")
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.map(lambda tu: _isa(tu, d), ["i2lqr_lane12", "i2lqr_group"])
    return {"lane12": a, "group": b}


def test_no_flat_memory_operations(isa):
    for tu, text in isa.items():
        n = len(re.findall(r"^\s+flat_(load|store|atomic)", text, flags=re.M))
        assert n == 0, f"{tu}: {n} flat_* operations (a pointer lost its address space)"


def test_hot_loops_of_the_quad12_lane_kernel_do_not_spill_to_scratch(isa):
    loops = _loops(isa["lane12"], "_ZN5i2lqr19k_lane_iterate_rowsIdNS_6Quad12IdEELb1EEE")
    # the Riccati step of the hot (branch-free) backward pass: the loop with the LDS-resident gains
    # and ~2000 fp64 operations; the forward / re-roll loops stream global memory without LDS
    hot = [ins for ins in loops.values()
           if 2500 < len(ins) < 4000 and sum(i.startswith("ds_") for i in ins) > 100]
    assert hot, {k: len(v) for k, v in loops.items()}
    step = min(hot, key=len)
    assert not [i for i in step if i.startswith("scratch_")]
    assert sum(i.startswith(("v_readlane", "v_writelane")) for i in step) < 150
    assert sum("f64" in i for i in step) > 1800
    # one full drain of the vector-memory counter per step at most: the landing of the next step's
    # inputs before the first gain store
    assert sum(i == "s_waitcnt" for i in step) < 60


def test_headline_kernel_loops_are_clean(isa):
    loops = _loops(isa["group"], "_ZN5i2lqr15k_group_iterateIdNS_8Bicycle6IdEELi3ELb0EEE")
    assert loops
    checked = 0
    for name, ins in loops.items():
        fp = sum("f64" in i for i in ins)
        if fp < 100 or len(ins) > 1200:
            continue  # not a horizon loop of the hot passes (the outer iteration loop, general forms)
        assert not [i for i in ins if i.startswith(("scratch_", "flat_"))], name
        assert not [i for i in ins if i.startswith(("global_", "buffer_"))], name  # state stays in LDS
        checked += 1
    assert checked >= 2  # the backward and the forward horizon loops at least
