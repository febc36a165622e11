"""Checks on the generated gfx950 code of librcw_hip (CPU only: hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_register_is_touched_between_an_asynchronous_load_and_its_wait():
    """The flat store kernel issues its descriptor loads one group ahead in inline asm (flat_load_*), so that the compiler
    does not drain the stores behind them when it waits; nothing but the hardware may then touch their destination
    registers until flat_wait_loads.  tools/check_async_loads.py proves that on the ISA: control-flow graph per kernel,
    forward data-flow of "load in flight" per register, every instruction checked."""
    csrc = os.path.join(ROOT, "raycastworlds.jl_amd", "csrc")
    res = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py"),
                          os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "asm", "rcw_kernels.s")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-4000:]
    assert "asynchronous loads checked" in res.stdout and not res.stdout.startswith("0 ")
