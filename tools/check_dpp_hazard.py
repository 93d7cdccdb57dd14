#!/usr/bin/env python3
"""DPP read-after-write hazard lint on a gfx950 ISA listing (run by `make` in csrc/ on the
translation unit that holds the sixteen-lane kernels, and by tests/test_isa_hygiene.py).

The row broadcasts of the sixteen-lane form are inline asm — the fp64 ALU's only DPP mode
(row_newbcast), folded into the multiply-add, is not something the compiler emits — so the
compiler's hazard recogniser does not see them: the hardware needs TWO wait states between a VALU
write of a VGPR and a DPP read of it.  For every *_dpp instruction the two issue slots in front of
it must not hold a VALU instruction whose destination overlaps the DPP source operand (an s_nop N
fills N + 1 slots).  Control flow is followed (a linear scan does not see a VALU write at
the end of a loop body in front of a DPP read at the loop head): where a block label sits in front
of the DPP instruction before two slots are accounted for, EVERY predecessor is walked — each branch
to that label and the fall-through from above — with the slots counted so far; a taken branch counts
as one slot (it costs far more).

    check_dpp_hazard.py listing.s [min_dpp_instructions]        exit 0 = clean
"""
import re
import sys


def vregs(tok):
    """VGPR numbers an operand token names: v12 -> {12}, v[4:7] -> {4..7}; anything else -> {}."""
    tok = tok.strip().rstrip(",").lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def check(text):
    """Returns (number of DPP instructions, list of violation strings)."""
    # instructions and labels in program order, per function (labels are local to a function)
    seq = []
    for l in text.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+|_Z\w+|[A-Za-z_]\w*):", l)
        if m:
            seq.append(("label", m.group(1)))
        elif l.startswith("\t") and not l.startswith(("\t;", "\t.")) and l.strip():
            seq.append(("ins", l))
    label_at = {name: i for i, (k, name) in enumerate(seq) if k == "label"}
    branches_to = {}
    for i, (k, l) in enumerate(seq):
        if k == "ins":
            q = l.split()
            if q[0].startswith(("s_cbranch", "s_branch")) and len(q) > 1:
                branches_to.setdefault(q[1], []).append(i)
    ENDS = ("s_branch", "s_endpgm", "s_setpc_b64")  # no fall-through behind these

    def walk(j, slots, src, dpp, seen):
        """All paths backward from position j (exclusive of the DPP instruction) until two issue
        slots are accounted for; yields violation strings."""
        while slots < 2 and j >= 0:
            kind, p = seq[j]
            if kind == "label":
                if p.startswith("_Z") or not p.startswith(".L"):
                    return  # function entry: nothing in front
                key = (p, slots)
                if key in seen:
                    return
                seen.add(key)
                # predecessors: every branch to this label, and the fall-through from above
                for bi in branches_to.get(p, []):
                    yield from walk(bi, slots, src, dpp, seen)
                k = j - 1
                while k >= 0 and seq[k][0] == "label":
                    k -= 1
                if k >= 0 and not seq[k][1].split()[0].startswith(ENDS):
                    yield from walk(k, slots, src, dpp, seen)
                return
            q = p.split()
            if q[0] == "s_nop":
                slots += int(q[1]) + 1
            else:
                slots += 1
                if q[0].startswith("v_") and len(q) > 1 and (vregs(q[1]) & src):
                    yield f"DPP hazard: {p.strip()!r} -> {dpp.strip()!r}"
            j -= 1

    found, bad = 0, []
    for i, (kind, l) in enumerate(seq):
        if kind != "ins":
            continue
        parts = l.split()
        if "_dpp" not in parts[0]:
            continue
        found += 1
        src = vregs(parts[2])  # v_fmac_*_dpp dst, SRC0 (the DPP operand), src1
        if not src:
            bad.append(f"unparsed DPP operand: {l.strip()!r}")
            continue
        bad.extend(walk(i - 1, 0, src, l, set()))
    return found, bad


if __name__ == "__main__":
    found, bad = check(open(sys.argv[1]).read())
    need = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for b in bad[:20]:
        print("check_dpp_hazard:", b, file=sys.stderr)
    if found < need:
        print(f"check_dpp_hazard: only {found} *_dpp instructions in {sys.argv[1]} (expected >= {need})",
              file=sys.stderr)
        sys.exit(2)
    print(f"check_dpp_hazard: {found} DPP instructions, {len(bad)} violations")
    sys.exit(1 if bad else 0)
