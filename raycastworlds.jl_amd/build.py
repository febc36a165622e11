"""Build librcw_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), f"-j{min(8, os.cpu_count() or 1)}"]   # (a translation unit per group of kernels: they compile side by side)
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building librcw_hip.so failed")
    return os.path.join(_PKG, "lib", "librcw_hip.so")
