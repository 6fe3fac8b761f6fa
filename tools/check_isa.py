"""Build-time guard against a register-allocator hazard seen with hipcc 7.2 on these kernels (DESIGN.md section 4 "Compiler hazard guard"; CHANGELOG.md rounds 2-3,
"Spills at divergent joins").

When a VGPR is spilled to an AGPR (or to scratch) at the top of the block where divergent control flow re-joins, the spill
store must come AFTER the `s_or_b64 exec, exec, ...` that re-enables the lanes which skipped the branch.  The compiler was
caught placing such stores BEFORE it: only the lanes of the branch save their value, the reload later runs for all lanes, and
the others read whatever the AGPR held.  The symptom is data-dependent garbage in lanes that did not take a branch -- or a
memory fault when the value is an address.  Which code is hit changes with every change of the register pressure, so the
check runs on the ISA of every build:

    python tools/check_isa.py <device assembly (.s) of fwsim.hip>

exits 1 and prints the blocks if a spill store to a register that is reloaded later sits between a block label and the exec
restore of that block.
"""
import re
import sys


def scan(text):
    hits = []
    for fn in re.split(r"\n(?=_Z[\w]+:\s)", text):
        name = fn.split(":", 1)[0]
        if not name.startswith("_Z"):
            continue
        lines = [l for l in fn.split("\n") if l.strip() and not re.match(r"\s*(\.loc|\.Ltmp|\.cfi|;|\.p2align)", l)]
        # MFMA accumulators live in AGPRs on purpose: a predicated v_accvgpr_write to one of them is an assignment, not a spill.
        # They are the AGPRs inside the operand ranges of the function's MFMAs; every other AGPR that is read back is a spill slot.
        acc = set()
        for ops in re.findall(r"v_mfma_\S+ ([^\n]*)", fn):
            for lo, hi in re.findall(r"a\[(\d+):(\d+)\]", ops):
                acc.update(f"a{i}" for i in range(int(lo), int(hi) + 1))
        reloaded = set(re.findall(r"v_accvgpr_read_b32 v\d+, (a\d+)", fn)) - acc
        for n, l in enumerate(lines):
            if not re.match(r"\.LBB\d+_\d+:", l):
                continue
            pre = []
            for x in lines[n + 1:n + 40]:
                if re.match(r"\.LBB", x) or "s_cbranch" in x or "s_branch" in x or "s_endpgm" in x:
                    break
                # another exec change first (the else-half of an if / else, a nested region): what follows is a new divergent
                # region whose stores are ordinary predicated assignments, not spills ahead of THIS block's restore
                if re.search(r"saveexec|s_xor_b64 exec|s_andn2_b64 exec|s_mov_b64 exec", x):
                    break
                if re.search(r"s_or_b64 exec, exec,", x):
                    spills = [p.strip() for p in pre
                              if (m := re.search(r"v_accvgpr_write_b32 (a\d+),", p)) and m.group(1) in reloaded
                              or re.search(r"scratch_store", p)]
                    # values staged in AGPRs as the data operand of a store of the same block are not spills
                    used_here = set(re.findall(r"a\[(\d+):(\d+)\]", " ".join(pre)))
                    staged = {f"a{i}" for lo, hi in used_here for i in range(int(lo), int(hi) + 1)}
                    spills = [p for p in spills if not ((m := re.search(r"v_accvgpr_write_b32 (a\d+),", p)) and m.group(1) in staged)]
                    if spills:
                        hits.append((name, l.split(":")[0], spills))
                    break
                pre.append(x)
    return hits


def scan_ticket(text, kernels=("fw_collect_stats_kernel",)):
    """Second guard: in the last-block-done kernels the ticket atomic must not be able to overtake the partial sums
    it announces.  The stores are write-through (sc1) and every wave drains them with an explicit `s_waitcnt vmcnt(0)` before
    the workgroup barrier behind which the ticket is taken.  Assert that shape in the ISA: walking up from the first
    `global_atomic_add` of the kernel there is an `s_barrier`, and walking up from that barrier a `vmcnt(0)` wait comes
    before any global store."""
    hits = []
    for fn in re.split(r"\n(?=_Z[\w]+:\s)", text):
        name = fn.split(":", 1)[0]
        if not name.startswith("_Z") or not any(k in name for k in kernels):
            continue
        lines = [l.strip() for l in fn.split("\n") if l.strip() and not re.match(r"\s*(\.loc|\.Ltmp|\.cfi|;|\.p2align)", l)]
        tick = next((i for i, l in enumerate(lines) if re.match(r"global_atomic_(add|inc)", l)), None)
        if tick is None:
            hits.append((name, "ticket", ["no ticket atomic found"]))
            continue
        bar = next((i for i in range(tick, -1, -1) if lines[i].startswith("s_barrier")), None)
        if bar is None:
            hits.append((name, "ticket", ["no s_barrier ahead of the ticket atomic"]))
            continue
        ok = False
        for i in range(bar - 1, -1, -1):
            if re.match(r"s_waitcnt\b.*vmcnt\(0\)", lines[i]):
                ok = True
                break
            if re.match(r"(global_store|global_atomic|buffer_store|flat_store)", lines[i]):
                break
        if not ok:
            hits.append((name, "ticket", ["a global store reaches the ticket barrier without s_waitcnt vmcnt(0)"]))
    return hits


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    h = scan(text) + scan_ticket(text)
    for name, label, spills in h:
        print(f"{name[:80]} {label}: spill stores before the exec restore: {spills[:6]}")
    print(f"{len(h)} suspicious block(s)")
    sys.exit(1 if h else 0)
