"""The top view's rule as data: rcw_api.hip keeps every threshold of update_top_view!'s choice of a form in ONE
table (kTopRules: name, value, unit, the profile line that put it there) and decides with a pure function of the configuration, the
batch and three numbers of the device (top_view_rule).  The development build exports both without needing a device:

  * the rule gives, for every shape of the committed profile table and for the batches its thresholds were measured at, exactly what
    tests/golden/top_view_plan_cases.json holds (written by tools/make_top_view_plan_cases.py) — an accidental change of a rule is a red
    test here, a deliberate retune is an edit of the table + a re-run of the tool;
  * for the profile's own shapes the forms are the ones the profile recorded its times with;
  * every rule names its evidence, and a profile it cites exists;
  * the device's numbers are arguments, not literals: another CU count scales the persistent grids, less LDS a CU or fewer wavefront
    slots change how many draw workgroups share a CU (and with it the number of parts an agent's fan goes to).
"""
import ctypes as C
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("make_top_view_plan_cases", os.path.join(ROOT, "tools", "make_top_view_plan_cases.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def devlib(rcw):
    from raycastworlds_jl_amd import _capi

    if not os.path.exists(_capi.DEV_LIB_PATH):
        from raycastworlds_jl_amd import build as _build

        _build.build()
    return _capi.load("dev")


def test_the_rule_gives_what_the_committed_cases_hold(devlib):
    tool = _tool()
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "top_view_plan_cases.json")))
    assert data["hw"] == tool.HW
    assert len(data["cases"]) >= 60
    for c in data["cases"]:
        got = tool.plan(devlib, c["H"], c["W"], c["pu"], c["N"], c["B"], c["Hc"])
        assert got == c["plan"], (c["source"], {k: c[k] for k in ("H", "W", "pu", "N", "B", "Hc")}, {k: (c["plan"][k], got[k]) for k in got if got[k] != c["plan"][k]})
    # the cases are the tool's: the profile table's shapes are all there
    name, shapes = tool.profile_shapes()
    assert len(shapes) >= 18
    have = {(c["H"], c["W"], c["pu"], c["N"], c["B"]) for c in data["cases"] if c["source"].startswith("profiles/")}
    assert {s[:5] for s in shapes} == have, name


def test_the_profile_table_was_taken_with_the_forms_the_rule_picks(devlib):
    tool = _tool()
    name, shapes = tool.profile_shapes()
    for H, W, pu, N, B, form, alone in shapes:
        got = tool.plan(devlib, H, W, pu, N, B)
        assert (got["form"], got["form_alone"]) == (form, alone), (name, H, W, pu, N, B)


def test_every_rule_names_its_evidence(devlib):
    buf = C.create_string_buffer(1 << 16)
    devlib.rcw_dev_top_view_rules.argtypes = [C.c_char_p, C.c_int32]
    n = devlib.rcw_dev_top_view_rules(buf, len(buf))
    assert n > 0
    rows = [l.split("\t") for l in buf.value.decode().strip().split("\n")]
    assert len(rows) == 26 and all(len(r) == 4 for r in rows)
    names = [r[0] for r in rows]
    assert len(set(names)) == len(names)
    for name, value, unit, evidence in rows:
        float(value)
        assert len(evidence) >= 10, name
        for f in re.findall(r"profiles/[\w.]+\.(?:txt|csv|json)", evidence):
            assert os.path.exists(os.path.join(ROOT, f)), (name, f)
    assert sum(1 for r in rows if "profiles/" in r[3]) >= 16               # (the others: an exactness limit, a rule's second factor, "same measurement")
    # the source holds no second copy of a threshold: the rule's function reads the table
    src = open(os.path.join(ROOT, "raycastworlds.jl_amd", "csrc", "rcw_api.hip")).read()
    body = src[src.index("int top_view_rule("):src.index("// Which form update_top_view! (SR:446-483) takes for this handle (top_view_rule)")]
    for literal in ("52 * 1024", "156 * 1024", ">= 65536", "24576", "12288", "6.5e6", "55.0", "34.0", "0.7 *", "160 * 1024", "<< 20"):
        assert literal not in body, literal


def test_the_devices_numbers_are_arguments(devlib):
    tool = _tool()
    base = tool.plan(devlib, 32, 32, 32, 1024, 64)                          # 1024^2 px x 64 agents on 256 CUs: four parts an agent
    assert base["top_parts"] == 4 and base["top_grid"] % 256 == 0
    hw = dict(tool.HW)
    try:
        tool.HW.update(cus=64)
        small = tool.plan(devlib, 32, 32, 32, 1024, 64)                     # 64 CUs: every CU has its agent, no parts
        assert small["top_parts"] == 1 and small["top_grid"] == base["top_grid"] // 4 and small["top_store_grid"] == 64
        tool.HW.update(hw); tool.HW.update(waves_per_cu=20)
        few = tool.plan(devlib, 24, 24, 32, 256, 114)                       # 768^2 px x 114: 16 wavefront slots left of 20 -> two draw workgroups of 512 threads a CU
        assert few["top_parts"] == 2
        tool.HW.update(hw); tool.HW.update(lds_per_cu=64 * 1024)
        lds = tool.plan(devlib, 8, 8, 32, 256, 4096)
        assert lds["form"] == "two-kernels"                                 # (the planes of a 256^2 px image are 8 KiB: any CU holds them)
    finally:
        tool.HW.clear(); tool.HW.update(hw)


def test_the_steps_rule_agrees_with_the_measured_crossovers(devlib):
    """rcw_api.hip, step_one_launch_pays: the one-launch step where the fill outlasts the casting half's five serial fans.  For every line of
    profiles/r06_small_batches.txt (us per step of both forms, four geometries, 1 .. 8192 agents) the rule picks the form that was measured
    faster — ties within 6 % (one box's run-to-run spread at these sizes) may go either way."""
    from raycastworlds_jl_amd import _capi

    geos = {"cfg1": (8, 8, 64), "cfg2": (8, 8, 256), "cfg3": (16, 16, 512), "cfg5": (32, 32, 1024)}
    devlib.rcw_dev_step_rule.argtypes = [C.POINTER(_capi.RcwConfig), C.c_int32, C.c_int32]
    text = open(os.path.join(ROOT, "profiles", "r06_small_batches.txt")).read()
    seen = wins = 0
    for m in re.finditer(r"^(?:(cfg\d) )?B=\s*(\d+).*?one-launch\s+([\d.]+) us/step \| two-launches\s+([\d.]+) us/step", text, re.M):
        geo, B, one, two = m.group(1) or "cfg2", int(m.group(2)), float(m.group(3)), float(m.group(4))
        H, W, N = geos[geo]
        cfg = _capi.default_config()
        cfg.height_tile_map_tu, cfg.width_tile_map_tu, cfg.num_rays = H, W, N
        rule = devlib.rcw_dev_step_rule(C.byref(cfg), B, 256)
        assert rule in (0, 1)
        seen += 1
        if abs(one - two) <= 0.06 * min(one, two):
            continue
        assert rule == (1 if one < two else 0), (geo, B, one, two, rule)
        wins += 1
    assert seen >= 25 and wins >= 20
    # what the geometry cannot take stays on two launches at any batch
    cfg = _capi.default_config(); cfg.height_camera_view_pu = 100          # (a camera height of the flat fill kernel)
    assert devlib.rcw_dev_step_rule(C.byref(cfg), 65536, 256) == 0
    cfg = _capi.default_config(); cfg.height_camera_view_pu = 128          # (the window kernels' heights take it)
    assert devlib.rcw_dev_step_rule(C.byref(cfg), 65536, 256) == 1 and devlib.rcw_dev_step_rule(C.byref(cfg), 64, 256) == 0
    cfg = _capi.default_config(); cfg.render_top_view = 1
    assert devlib.rcw_dev_step_rule(C.byref(cfg), 65536, 256) == 0


@pytest.mark.gpu
def test_a_handle_takes_the_form_the_rule_gives_for_its_device(devlib, rcw):
    torch = pytest.importorskip("torch")
    tool = _tool()
    p = torch.cuda.get_device_properties(0)
    hw = dict(tool.HW)
    try:
        tool.HW.update(cus=p.multi_processor_count, lds_per_cu=p.shared_memory_per_block, waves_per_cu=p.max_threads_per_multi_processor // 64)
        for H, W, pu, N, B in ((8, 8, 32, 256, 64), (8, 8, 10, 256, 4096), (8, 8, 20, 256, 512), (24, 24, 32, 256, 114), (8, 8, 5, 64, 16), (8, 8, 200, 64, 4)):
            want = tool.plan(devlib, H, W, pu, N, B)
            with rcw.SingleRoomModule.SingleRoom(batch=B, seed=1, height_tile_map_tu=H, width_tile_map_tu=W, pu_per_tu=pu, num_rays=N, render_top_view=True) as env:
                assert (env.top_view_form(), env.update_top_view_form()) == (want["form"], want["form_alone"]), (H, W, pu, N, B)
    finally:
        tool.HW.clear(); tool.HW.update(hw)
