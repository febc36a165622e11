"""The C-ABI library loads and exports every symbol include/rcw.h declares; without a GPU it
fails loudly instead of falling back to anything.  CPU only — no compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rcw.h")


def _declared():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"RCW_API\s+[\w\s\*]+?\b(rcw_\w+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("rcw_create", "rcw_destroy", "rcw_reset", "rcw_set_state", "rcw_step", "rcw_step_device",
                 "rcw_obs_device_ptr", "rcw_reward", "rcw_done", "rcw_tile_map_chunks", "rcw_rays",
                 "rcw_columns", "rcw_expand_columns", "rcw_last_error"):
        assert must in names
    assert len(names) >= 35


def test_library_exports_every_declared_symbol(rcw):
    from raycastworlds_jl_amd import _capi

    lib = C.CDLL(_capi.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"declared in rcw.h but not exported: {missing}"
    # and the Python binding binds exactly that set
    assert sorted(_capi.SIGNATURES) == _declared()


def test_config_struct_matches_header(rcw, oracle):
    from raycastworlds_jl_amd import _capi

    cfg = _capi.default_config()          # rcw_config_default: host-only, no device touched
    assert C.sizeof(_capi.RcwConfig) == C.sizeof(oracle.RcwConfig) == 160
    ref = oracle.default_config()
    for name, _ in _capi.RcwConfig._fields_:
        if name == "reserved":
            continue
        assert getattr(cfg, name) == getattr(ref, name), name
    # reference defaults SR:258-272, SR:288-296
    assert (cfg.height_tile_map_tu, cfg.width_tile_map_tu, cfg.num_directions, cfg.num_rays,
            cfg.height_camera_view_pu, cfg.pu_per_tu) == (8, 16, 128, 512, 256, 32)
    assert cfg.player_radius_wu == 0.125 and cfg.position_increment_wu == 0.125
    assert (cfg.floor_color, cfg.ceiling_color, cfg.wall_dim_1_color, cfg.wall_dim_2_color,
            cfg.goal_dim_1_color, cfg.goal_dim_2_color) == (0x404040, 0xFFFFFF, 0x808080, 0xC0C0C0, 0x800000, 0xC00000)


def test_no_cpu_fallback(rcw):
    """On a machine without a gfx950 device construction fails loudly (RCW_ERR_NO_DEVICE)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raycastworlds_jl_amd import _capi

    with pytest.raises(_capi.RcwError) as ei:
        rcw.SingleRoomModule.SingleRoom(batch=2)
    assert ei.value.code == _capi.RCW_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    """The product package must not reference oracle/ (it is the checker, not a fallback)."""
    pkg = os.path.join(ROOT, "raycastworlds.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in text.lower(), (dirpath, fn)


def test_shipped_library_reads_no_development_switch(rcw):
    """The shipped librcw_hip.so names exactly one environment variable, RCW_RCCL_LIBRARY (where to find RCCL): the
    tuning knobs and measured-and-rejected kernel variants of development live in librcw_hip_dev.so only, so a stray
    RCW_* variable in a user's shell cannot change results or performance."""
    from raycastworlds_jl_amd import _capi

    def env_names(path):
        data = open(path, "rb").read()
        return sorted(set(m.decode() for m in re.findall(rb"(?<![A-Z_0-9])RCW_[A-Z0-9_]{3,}(?=\x00)", data)))

    assert env_names(_capi.LIB_PATH) == ["RCW_RCCL_LIBRARY"]
    # ... nor does it carry the development-only kernels (the round-3 cast kernel and its variants, the wavefront-per-agent form)
    shipped = open(_capi.LIB_PATH, "rb").read()
    assert b"rcw_cast_kernel_r3" not in shipped and b"rcw_cast_waves_kernel" not in shipped and b"rcw_step256_kernel" not in shipped
    assert b"rcw_cast_kernel" in shipped and b"rcw_fill256_draw_kernel" in shipped
    if os.path.exists(_capi.DEV_LIB_PATH):
        dev = env_names(_capi.DEV_LIB_PATH)
        assert "RCW_CAST_MARCH" in dev and "RCW_TOP_DEBUG" in dev and "RCW_RCCL_LIBRARY" in dev
        assert "RCW_CAST_KERNEL" in dev and "RCW_TOP_FUSED" in dev and "RCW_STEP_FUSED" in dev
        assert b"rcw_cast_kernel_r3" in open(_capi.DEV_LIB_PATH, "rb").read() and b"rcw_step256_kernel" in open(_capi.DEV_LIB_PATH, "rb").read()
        # the development build exports the same ABI
        lib = C.CDLL(_capi.DEV_LIB_PATH)
        assert not [n for n in _declared() if not hasattr(lib, n)]


def test_every_entry_point_refuses_a_null_handle(rcw):
    """No entry point dereferences a NULL handle: each returns a negative RCW_ERR_* with a message (rcw_destroy(NULL) is a
    no-op, like free).  Runs without a GPU: the check comes before any HIP call."""
    from raycastworlds_jl_amd import _capi

    lib = _capi.load()
    called = 0
    for name, sig in _capi.SIGNATURES.items():
        if not sig or sig[0] is not C.c_void_p or name in ("rcw_destroy",):
            continue
        args = [None]
        for t in sig[1:]:
            if t in (C.c_float, C.c_double):
                args.append(0.0)
            elif t is C.c_void_p or hasattr(t, "contents"):
                args.append(None)
            else:
                args.append(t(0))
        rc = getattr(lib, name)(*args)
        assert isinstance(rc, int) and rc < 0, (name, rc)
        assert _capi.last_error(lib), name
        called += 1
    assert called >= 50
    assert lib.rcw_destroy(None) == 0
