"""Static check of a hand-scheduled gfx950 loop: replay the compiled loop body against a model of the wave's in-order VMEM
queue and report every instruction that touches a register a load may still be writing, and every barrier that an LDS-DMA
(global_load_lds) may still be in flight across.

hipcc does not count inline-asm memory operations, so the Winograd conv kernel (csrc/i2v_conv16w.hip) issues its loads as
asm and writes every `s_waitcnt vmcnt(n)` by hand; this script is the independent check of that arithmetic on the code the
compiler actually produced (tests/test_host_cpu.py runs it on every instantiation).

    python tools/check_asm_waits.py file.s [kernel-name-regex]
"""
import re
import sys

REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def regs(operand_text):
    out = set()
    for a, b, c in REG.findall(operand_text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def kernel_loops(text, name_re):
    """(kernel name, loop body lines) of every kernel matching name_re: the loop = the backward branch that holds MFMAs."""
    for name, body in re.findall(r"^(\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M):
        if not re.search(name_re, name):
            continue
        loops = [m.group(2) for m in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n((?:(?!^\.LBB).)*?)s_cbranch_\w+ \1\n", body,
                                                 flags=re.S | re.M) if "v_mfma" in m.group(2)]
        yield name, loops


def check_loop_entries(text, name_re):
    """The replay below starts every loop with an EMPTY queue, and the hand-written wait counts assume the steady state of the
    loop: both are only right if nothing is in flight when the loop is entered.  So: walking back from the loop header, a
    `s_waitcnt vmcnt(0)` must come before any load.  (Rounds 2-3 shipped loops whose prologue left its weight requests in
    flight; the first taps then under-waited -- wrong results once in a few hundred launches of one shape.)"""
    bad = []
    for name, body in re.findall(r"^(\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M):
        if not re.search(name_re, name):
            continue
        for m in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n((?:(?!^\.LBB).)*?)s_cbranch_\w+ \1\n", body, flags=re.S | re.M):
            if "v_mfma" not in m.group(2):
                continue
            before = [l.split(";")[0].strip() for l in body[:m.start()].split("\n")]
            verdict = "no s_waitcnt vmcnt(0) in front of the loop"
            for l in reversed(before):
                if re.match(r"s_waitcnt\b.*vmcnt\(0\)", l):
                    verdict = None
                    break
                if l.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                    verdict = f"'{l}' is issued after the last vmcnt(0) in front of the loop"
                    break
            if verdict:
                bad.append(f"{name} loop at {m.group(1)}: {verdict}")
    return bad


def check_loop(loop_text, iterations=3):
    """Returns a list of violation strings (empty = the schedule is safe under in-order VMEM return)."""
    lines = [l.split(";")[0].strip() for l in loop_text.split("\n")]
    lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
    queue = []          # in-flight VMEM operations, oldest first: ("reg", {dest regs}) or ("lds", None)
    bad = []
    for it in range(iterations):
        for ln, l in enumerate(lines):
            op, _, rest = l.partition(" ")
            pending = set().union(*[q[1] for q in queue if q[0] == "reg"]) if queue else set()
            if op.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", l)
                if m:
                    n = int(m.group(1))
                    del queue[:max(0, len(queue) - n)]
                continue
            if op == "s_barrier":
                if any(q[0] == "lds" for q in queue):
                    bad.append(f"iteration {it}, line {ln}: s_barrier with an LDS-DMA load possibly in flight")
                continue
            if op.startswith(("global_store", "buffer_store", "scratch_", "flat_store", "global_atomic")):
                bad.append(f"iteration {it}, line {ln}: '{l}' -- an uncounted VMEM operation inside the loop")
                continue
            touched = regs(rest)
            if touched & pending:
                bad.append(f"iteration {it}, line {ln}: '{l}' touches v{sorted(touched & pending)} while a load may still write them")
            if op.startswith("global_load_lds") or (op.startswith("buffer_load") and re.search(r"\blds\b", rest)):
                queue.append(("lds", None))     # LDS-DMA: no VGPR destination (a buffer form's first operand is the OFFSET register)
            elif op.startswith(("global_load", "buffer_load", "flat_load")):
                queue.append(("reg", regs(rest.split(",")[0])))
    return bad


def check_scalar_operands(text, name_re):
    """The asm loads that take their address from an SGPR pair: the compiler's hazard recogniser does not look inside inline
    asm, so a VALU instruction (v_readfirstlane, v_readlane, v_cmp) that writes such an SGPR must not sit within five
    instructions in front of the load (gfx9: "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states")."""
    bad = []
    for name, body in re.findall(r"^(\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M):
        if not re.search(name_re, name):
            continue
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
        for i, l in enumerate(lines):
            if not l.startswith(("global_load", "buffer_load")):
                continue
            sregs = set()
            for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", l):
                sregs.update(range(int(a), int(b) + 1))
            for k in range(max(0, i - 5), i):
                m = re.match(r"(v_readfirstlane\w*|v_readlane\w*|v_cmp\w*)\s+s(?:\[(\d+):(\d+)\]|(\d+))", lines[k])
                if m:
                    w = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
                    if w & sregs:
                        bad.append(f"{name}: '{lines[k]}' {i - k} instructions in front of '{l}'")
    return bad


def main():
    text = open(sys.argv[1]).read()
    name_re = sys.argv[2] if len(sys.argv) > 2 else "."
    rc = 0
    for name, loops in kernel_loops(text, name_re):
        for i, loop in enumerate(loops):
            bad = check_loop(loop)
            print(f"{name} loop {i}: {'ok' if not bad else str(len(bad)) + ' violations'}")
            for b in bad[:20]:
                print("   ", b)
            rc |= bool(bad)
    for e in check_loop_entries(text, name_re):
        print("loop entry:", e)
        rc = 1
    hz = check_scalar_operands(text, name_re)
    for h in hz[:20]:
        print("SGPR hazard:", h)
    rc |= bool(hz)
    return rc


if __name__ == "__main__":
    sys.exit(main())
