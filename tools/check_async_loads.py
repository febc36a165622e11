#!/usr/bin/env python3
"""Build check for the store kernels' descriptor prefetch (flat_load_* / flat_wait_loads in rcw_top_store.hip).

Those global loads are issued in inline asm so that the compiler does not track their completion; the price is that
nothing but the hardware may touch a destination register between the load and the `s_waitcnt vmcnt` that awaits it.
This script reads the generated ISA (make asm -> lib/asm/rcw_kernels.s), builds every kernel's control-flow graph and
runs a forward data-flow analysis: the set of registers with a load in flight at each instruction (union over all
paths; born at an inline-asm `global_load_*`, killed by any `s_waitcnt vmcnt(..)`).  It fails if an instruction names
a register while a load into it may be in flight — a copy the register allocator slipped in, a use that moved up.
(The count inside vmcnt(N) is not modelled: flat_wait_loads<63> is placed where exactly 64 stores follow the loads.)
"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "raycastworlds.jl_amd/lib/asm/rcw_kernels.s"


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def analyse(name, lines):
    # instructions: (text, in_asm); labels -> instruction index
    ins, labels, local, in_asm = [], {}, {}, False
    for t in lines:
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins); continue
        m = re.match(r"^(\d+):$", t)                      # a numeric local label of an asm statement ("1:", branched to as 1f / 1b)
        if m:
            local.setdefault(m.group(1), []).append(len(ins)); continue
        if not t or t.startswith((";", ".", "//")):
            continue
        ins.append((t, in_asm))
    n = len(ins)
    succ = [[] for _ in range(n)]

    def target(i, name):
        m = re.match(r"^(\d+)([fb])$", name)
        if not m:
            return labels[name]
        at = local[m.group(1)]
        return min(x for x in at if x > i) if m.group(2) == "f" else max(x for x in at if x <= i)
    for i, (t, _) in enumerate(ins):
        op = t.split()[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(target(i, t.split()[1])); continue
        if op.startswith("s_cbranch"):
            succ[i].append(target(i, t.split()[1]))
        if i + 1 < n:
            succ[i].append(i + 1)
    state = [None] * n          # pending registers BEFORE instruction i
    state[0] = frozenset()
    work, bad, loads = [0], {}, 0

    def transfer(i, pend):
        t, asm = ins[i]
        op = t.split()[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if op.startswith("s_waitcnt") and "vmcnt" in t:
            return frozenset(), set()
        if asm and op.startswith("global_load"):
            used = set().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
            return pend | regs(toks[0]), used & pend
        used = set().union(*[regs(x) for x in toks]) if toks else set()
        return pend, used & pend

    while work:
        i = work.pop()
        out, hit = transfer(i, state[i])
        if hit:
            bad[i] = sorted(hit)
        for j in succ[i]:
            new = out if state[j] is None else state[j] | out
            if new != state[j]:
                state[j] = new
                work.append(j)
    loads = sum(1 for t, a in ins if a and t.startswith("global_load"))
    return loads, [(ins[i][0], h) for i, h in sorted(bad.items())]


kernels, cur, name = {}, None, None
for line in open(path):
    t = line.strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
        name, cur = m.group(1), []
        kernels[name] = cur
        continue
    if cur is not None:
        if t.startswith(".Lfunc_end"):
            cur = None
        else:
            cur.append(t)
total, failed = 0, False
for name, lines in kernels.items():
    if "global_load" not in "\n".join(lines) or ";;#ASMSTART" not in "\n".join(lines):
        continue
    loads, bad = analyse(name, lines)
    if not loads:
        continue
    total += loads
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:48]
    print(f"{short:48s} {loads:3d} asynchronous loads, {len(bad)} hazards")
    for t, h in bad[:10]:
        failed = True
        print(f"    HAZARD `{t}` touches v{h} while a load into it may be in flight")
print(f"{total} asynchronous loads checked")
sys.exit(1 if failed else 0)
