#!/usr/bin/env python3
"""Dev tool: compare the generated gfx950 ISA of every kernel between two listings made by `make asm`
(lib/asm/*.s), kernel by kernel — for refactorings that must not change the shipped kernels' code.

    python tools/isa_diff.py OLD.s NEW.s [--only SUBSTR] [--show]
    python tools/isa_diff.py OLD.s NEW_DIR        (a directory: every *.s in it, concatenated)

A kernel's body is the text between its `<symbol>:` label and its `.Lfunc_end`; comments, debug directives and
local label NUMBERS (which shift when a translation unit is cut differently) are normalised away.  Exit code 1 when
a kernel present in both differs; kernels only on one side are listed."""
from __future__ import annotations

import argparse
import difflib
import glob
import os
import re
import sys


def read(path: str) -> str:
    if os.path.isdir(path):
        return "\n".join(open(f).read() for f in sorted(glob.glob(os.path.join(path, "*.s"))))
    return open(path).read()


def kernels(text: str) -> dict[str, list[str]]:
    out: dict[str, list[str]] = {}
    name, body = None, []
    kernel_syms = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M))
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and m.group(1) in kernel_syms:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = normalise(body)
            name = None
            continue
        body.append(line)
    return out


def normalise(body: list[str]) -> list[str]:
    labels: dict[str, str] = {}
    res = []
    for line in body:
        line = line.split(";")[0].rstrip()
        s = line.strip()
        if not s or s.startswith((".loc", ".file", ".cfi", ".p2align", ".Ltmp", "s_nop", "s_code_end")):
            continue
        res.append(s)
    # local labels: renumber in order of appearance
    def lab(m):
        k = m.group(0)
        if k not in labels:
            labels[k] = f".L{len(labels)}"
        return labels[k]
    return [re.sub(r"\.LBB\d+_\d+", lab, s) for s in res]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("old"); ap.add_argument("new")
    ap.add_argument("--only", default="")
    ap.add_argument("--show", action="store_true")
    a = ap.parse_args()
    ko, kn = kernels(read(a.old)), kernels(read(a.new))
    names = sorted(set(ko) | set(kn))
    same = diff = 0
    for n in names:
        if a.only and a.only not in n:
            continue
        if n not in ko:
            print(f"only in NEW: {n}"); continue
        if n not in kn:
            print(f"only in OLD: {n}"); continue
        if ko[n] == kn[n]:
            same += 1
        else:
            diff += 1
            print(f"DIFFERS ({len(ko[n])} -> {len(kn[n])} lines): {n}")
            if a.show:
                for l in list(difflib.unified_diff(ko[n], kn[n], lineterm="", n=2))[:80]:
                    print("    " + l)
    print(f"{same} kernels identical, {diff} differ ({len(ko)} in OLD, {len(kn)} in NEW)")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
