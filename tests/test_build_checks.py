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


SYNTHETIC = """\
_Z{name}:
	v_mov_b32_e32 v1, 0
	;;#ASMSTART
	global_load_dword v5, v[2:3], off
	;;#ASMEND
{between}
.LBB0_1:
	global_store_dwordx4 v[8:9], v[10:13], off nt
	s_cbranch_scc1 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(63)
	;;#ASMEND
	v_add_u32_e32 v6, v5, v5
	s_endpgm
.Lfunc_end0:
"""


def _check(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py"), str(f)], capture_output=True, text=True)


def test_the_checker_sees_a_hazard_when_there_is_one(tmp_path):
    """The guard itself: a clean kernel passes; a register copy slipped in between the load and its wait, a use on one
    branch only, and a use inside the store loop are each reported."""
    ok = _check(tmp_path, SYNTHETIC.format(name="clean", between="\tv_add_u32_e32 v7, v1, v1"))
    assert ok.returncode == 0 and "1 asynchronous loads, 0 hazards" in ok.stdout, ok.stdout
    copy = _check(tmp_path, SYNTHETIC.format(name="copy", between="\tv_mov_b32_e32 v9, v5"))
    assert copy.returncode == 1 and "HAZARD" in copy.stdout and "v_mov_b32_e32 v9, v5" in copy.stdout, copy.stdout
    branch = _check(tmp_path, SYNTHETIC.format(name="branch", between="\ts_cbranch_scc0 .LBB0_1\n\tv_add_u32_e32 v7, v5, v1"))
    assert branch.returncode == 1 and "HAZARD" in branch.stdout, branch.stdout
    in_loop = _check(tmp_path, SYNTHETIC.format(name="loop", between="").replace("global_store_dwordx4 v[8:9], v[10:13]", "global_store_dwordx4 v[4:5], v[10:13]"))
    assert in_loop.returncode == 1 and "HAZARD" in in_loop.stdout and "global_store_dwordx4 v[4:5]" in in_loop.stdout, in_loop.stdout


def test_the_fill_kernels_prefetch_keeps_its_three_dependent_round_trips():
    """rcw_fill256_kernel's descriptor prefetch is part of its pace (docs/experiments.md): height -> colour id -> colour, each
    load awaited before the next is issued.  Every shorter form measured makes the kernel slower, the more so the larger the
    batch — round 4 lost 13 % at 8 GiB when a refactoring let the compiler issue the first two loads together.  Checked on the
    ISA of the kernel proper and of rcw_fill256_draw_kernel (the same body beside the top view's drawing)."""
    import re

    csrc = os.path.join(ROOT, "raycastworlds.jl_amd", "csrc")
    res = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    text = open(os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "asm", "rcw_kernels.s")).read()
    kernels = re.findall(r"^(_ZN12_GLOBAL__N_1\d+rcw_fill256_(?:draw_)?kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) == 1 + 8, [k for k, _ in kernels]            # the kernel proper (non-temporal stores; <PLAIN> is the development build's), the fused kernel x (T, TIE_LE, DIST_PRE)
    for name, body in kernels:
        ops = [l.strip() for l in body.splitlines() if re.match(r"\s+(global_load|s_waitcnt vmcnt\(0\))", l)]
        # the three descriptor loads, in order, with a full wait between each pair
        i_h = next(i for i, o in enumerate(ops) if o.startswith("global_load_dword") and "off" in o and "s[0:1]" not in o)
        i_c = next(i for i, o in enumerate(ops) if i > i_h and o.startswith("global_load_ubyte"))
        i_k = next(i for i, o in enumerate(ops) if i > i_c and o.startswith("global_load_dword") and "s[0:1]" in o)
        assert any(o.startswith("s_waitcnt vmcnt(0)") for o in ops[i_h + 1:i_c]), f"{name}: height and colour id are requested together"
        assert any(o.startswith("s_waitcnt vmcnt(0)") for o in ops[i_c + 1:i_k]), f"{name}: colour id and colour are requested together"


def test_the_one_launch_steps_chunk_loops_start_on_a_line(rcw):
    """rcw_fill256_cast_kernel / rcw_fill_window_cast_kernel hold the fill's chunk loop and the casting half in one function; in builds where the chunk loop
    started 4 or 36 bytes into a 64-byte line of the code, the launch was 25-30 us (2 %) slower at 16384 x 512 and 8192 x 1024 view columns than in builds
    where it started at 0 or 28 — whatever the casting code in front of it had been changed by (profiles/r06_step_forms.txt (5)).  csrc/Makefile compiles
    rcw_cast.hip (and rcw_top_draw.hip, whose rcw_fill256_draw_kernel is the same loop beside the top view's drawing) with -falign-loops=64; this reads the shipped library's code objects (tools/loop_lines.py) and wants every chunk loop of every one of the 32
    instantiations at byte 0 of a line."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import loop_lines

    from raycastworlds_jl_amd import _capi

    lib = _capi.LIB_PATH                                                 # (the `rcw` fixture has built it if this checkout had none)
    seen = 0
    for family, sizes in (("rcw_fill256_cast_kernel", 2), ("rcw_fill_window_cast_kernel", 3)):
        per_kernel = loop_lines.chunk_loops(lib, family)
        assert len(per_kernel) == 16, (family, sorted(per_kernel))
        for name, chunk in per_kernel.items():
            assert len(chunk) >= sizes, (name, chunk)                # (256 rows: the loop unrolled by four and its tail; the window: a body per M)
            assert all(start == 0 for _, start in chunk), (name, chunk)
            seen += len(chunk)
    assert seen >= 16 * 5
    # ... and rcw_fill256_draw_kernel (rcw_top_draw.hip, the same flag): the camera fill's chunk loop beside the top view's drawing
    per_kernel = loop_lines.chunk_loops(lib, "rcw_fill256_draw_kernel")
    assert len(per_kernel) == 8, sorted(per_kernel)
    for name, chunk in per_kernel.items():
        fill = [c for c in chunk if c[0] in (540, 684)]                  # (the other loops with 16-byte stores copy the bit plane out)
        assert len(fill) == 2 and all(start == 0 for _, start in fill), (name, chunk)
