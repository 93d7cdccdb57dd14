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

# The COUNT thresholds below (loop sizes, lane moves, waits) were calibrated on this compiler; on
# another one they are reported as expected failures, the hard invariants (no flat_* / scratch_* in
# hot loops, no DPP read-after-write hazard) are asserted on any.
CALIBRATED_HIPCC = "7.2.26015"

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
    start = next((i for i, l in enumerate(lines) if l.startswith(kernel_prefix) and ":" in l), None)
    assert start is not None, (f"no kernel symbol starting with {kernel_prefix} in the ISA listing: "
                               "the kernel was renamed or its template arguments changed")
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
    with ThreadPoolExecutor(3) as ex:
        a, b, c = ex.map(lambda tu: _isa(tu, d), ["i2lqr_lane12", "i2lqr_group", "i2lqr_lanepair"])
    return {"lane12": a, "group": b, "lanepair": c}


def _counts_apply():
    """True on the compiler the count thresholds were calibrated on; otherwise the caller turns a
    threshold miss into an expected failure."""
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:  # noqa: BLE001
        return False
    return CALIBRATED_HIPCC in out


def _soft(cond, msg):
    if cond:
        return
    if _counts_apply():
        raise AssertionError(msg)
    pytest.xfail(f"{msg} (count threshold calibrated on hipcc {CALIBRATED_HIPCC}, another compiler here)")


def test_no_flat_memory_operations(isa):
    for tu, text in isa.items():
        n = len(re.findall(r"^\s+flat_(load|store|atomic)", text, flags=re.M))
        assert n == 0, f"{tu}: {n} flat_* operations (a pointer lost its address space)"


def test_hot_loops_of_the_quad12_lane_kernel_do_not_spill_to_scratch(isa):
    loops = _loops(isa["lane12"], "_ZN5i2lqr19k_lane_iterate_rowsIdNS_6Quad12IdEELb0ELb1EEE")
    # the Riccati step of the hot (branch-free) backward pass: the loop with the LDS-resident gains
    # and ~2000 fp64 operations; the forward / re-roll loops stream global memory without LDS
    hot = [ins for ins in loops.values()
           if 2500 < len(ins) < 4000 and sum(i.startswith("ds_") for i in ins) > 100]
    assert hot, {k: len(v) for k, v in loops.items()}
    step = min(hot, key=len)
    assert not [i for i in step if i.startswith("scratch_")]
    _soft(sum(i.startswith(("v_readlane", "v_writelane")) for i in step) < 150, "lane moves in the step")
    _soft(sum("f64" in i for i in step) > 1800, "fp64 operations in the step")
    # one full drain of the vector-memory counter per step at most: the landing of the next step's
    # inputs before the first gain store
    _soft(sum(i == "s_waitcnt" for i in step) < 60, "s_waitcnt in the step")


def test_headline_kernel_loops_are_clean(isa):
    # the headline launch: sixteen lanes per problem, one helper wavefront (H = 2, WS = false, G = 16)
    loops = _loops(isa["group"], "_ZN5i2lqr15k_group_iterateIdNS_8Bicycle6IdEELi2ELb0ELi16EEE")
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


def test_helper_wavefront_kernel_hands_its_record_over_in_lds_and_spills_to_registers_only(isa):
    """k_lane_iterate_pair (round 5): the per-step barrier of the pair orders LDS traffic only
    (s_waitcnt lgkmcnt(0) + s_barrier from the inline asm: no vmcnt drain in front of it), and the
    horizon loops of both wavefronts keep their spills in accumulation registers, not in scratch."""
    text = isa["lanepair"]
    loops = _loops(text, "_ZN5i2lqr19k_lane_iterate_pairIdNS_8Bicycle6IdEELb0ELb1EEE")
    assert loops
    hot = {k: v for k, v in loops.items() if sum("f64" in i for i in v) > 100}
    assert len(hot) >= 4, {k: len(v) for k, v in loops.items()}  # helper, main, forward, re-roll
    for name, ins in hot.items():
        assert not [i for i in ins if i.startswith(("scratch_", "flat_"))], name
    lines = [l.strip() for l in text.split("\n")]
    bars = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
    assert bars
    asm_bars = [i for i in bars if lines[i - 1].startswith("s_waitcnt lgkmcnt(0)")]
    assert len(asm_bars) >= 2, "the pair's LDS-only barrier is gone from the backward passes"


def _lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "check_dpp_hazard", Path(__file__).resolve().parent.parent / "tools" / "check_dpp_hazard.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_dpp_read_of_a_register_written_by_the_two_instructions_in_front(isa):
    """The DPP row broadcasts of the sixteen-lane form are inline asm (the fp64 ALU's only DPP mode,
    folded into the multiply-add, is not something the compiler emits), so its hazard recogniser
    does not see them: the hardware needs two wait states between a VALU write of a VGPR and a DPP
    read of it.  Checked on the compiled code by tools/check_dpp_hazard.py — the same check `make`
    runs before it links the library — along every control-flow predecessor of a DPP instruction."""
    found, bad = _lint().check(isa["group"])
    assert not bad, bad[:5]
    assert found > 100  # the sixteen-lane kernels are in this translation unit


def test_the_dpp_lint_sees_hazards_across_block_boundaries():
    """a linear scan misses a VALU write at the end of a loop body in front of a DPP read
    at the loop head.  Synthetic listings: a hazard over a back-edge, over a branch into a block,
    the same with a wait state inside the block (clean), and a straight-line one."""
    chk = _lint().check
    dpp = "\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    back_edge = ("_Z1kv:\n\tv_mov_b32_e32 v9, 0\n.LBB0_1:\n" + dpp +
                 "\tv_add_f64 v[6:7], v[6:7], v[6:7]\n\tv_mul_f64 v[2:3], v[6:7], v[6:7]\n"
                 "\ts_cbranch_vccnz .LBB0_1\n\ts_endpgm\n")
    assert chk(back_edge)[1], "write of v[2:3] one slot (the branch) in front of the loop-head DPP read"
    clean = back_edge.replace(".LBB0_1:\n" + dpp, ".LBB0_1:\n\ts_nop 1\n" + dpp)
    assert chk(clean) == (1, [])
    branch_in = ("_Z1kv:\n\tv_mul_f64 v[2:3], v[6:7], v[6:7]\n\ts_cbranch_vccz .LBB0_2\n"
                 "\tv_mov_b32_e32 v9, 0\n\tv_mov_b32_e32 v9, 0\n.LBB0_2:\n" + dpp + "\ts_endpgm\n")
    assert len(chk(branch_in)[1]) == 1  # over the taken branch only; the fall-through has two slots
    straight = "_Z1kv:\n\tv_mul_f64 v[2:3], v[6:7], v[6:7]\n\tv_mov_b32_e32 v9, 0\n" + dpp
    assert chk(straight)[1]
    assert chk(straight.replace("\tv_mov_b32_e32 v9, 0\n", "\ts_nop 1\n")) == (1, [])
