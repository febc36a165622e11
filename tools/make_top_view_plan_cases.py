#!/usr/bin/env python3
"""tests/golden/top_view_plan_cases.json: what the top view's rule (rcw_api.hip: kTopRules, top_view_rule) decides for every shape the
committed profiles were taken with — the table of profiles/*_top_view_shapes.txt (~1 GiB of top view a launch) and the batches the rule's
thresholds were measured at (kTopRules' evidence: small batches, big images in few / many agents, another camera height).

    python tools/make_top_view_plan_cases.py            # rewrite the file from the development build's rcw_dev_plan_top_view (CPU)

Run it after a DELIBERATE retune (an edit of kTopRules + a re-run of tools/top_view_shapes.py on the GPU box); tests/test_top_view_plan.py
compares the rule with the file on every CPU run and cross-checks the forms with the ones the profile table recorded."""
import ctypes as C
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raycastworlds_jl_amd import _capi  # noqa: E402

HW = dict(cus=256, lds_per_cu=160 * 1024, waves_per_cu=32)      # an MI355X in SPX mode (hipDeviceProp_t, tools/_build/props.py on the box)
FIELDS = ("form", "form_alone", "top_lds", "top_split", "top_flat", "top_unit_px", "top_fused", "top_draw_first", "top_parts", "top_runs",
          "top_draw_block", "top_draw_block_alone", "top_alone_split", "top_grid", "top_store_grid", "rc")
FORMS = ("none", "in-place", "one-kernel", "two-kernels")


def plan(lib, H, W, pu, N, B, Hc=256, want_form=0, want_runs=0, **kw):
    cfg = _capi.default_config()
    cfg.height_tile_map_tu, cfg.width_tile_map_tu, cfg.pu_per_tu, cfg.num_rays, cfg.height_camera_view_pu, cfg.render_top_view = H, W, pu, N, Hc, 1
    for k, v in kw.items():
        setattr(cfg, k, v)
    out = (C.c_int32 * 16)()
    lib.rcw_dev_plan_top_view.argtypes = [C.POINTER(_capi.RcwConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.rcw_dev_plan_top_view(C.byref(cfg), B, HW["cus"], HW["lds_per_cu"], HW["waves_per_cu"], want_form, want_runs, out)
    d = dict(zip(FIELDS, [int(v) for v in out]))
    d["form"], d["form_alone"] = FORMS[d["form"]], FORMS[d["form_alone"]]
    return d


def profile_shapes():
    """(H, W, pu, N, B, form in a step, form alone) of every shape line of the newest committed shapes table"""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_top_view_shapes\.txt", f))
    text = open(os.path.join(ROOT, "profiles", files[-1])).read()
    out = []
    for m in re.finditer(r"^map\s*(\d+)x\s*(\d+) pu\s*(\d+) N\s*(\d+) image\s*\d+x\s*\d+ B\s*(\d+) ([\w-]+): .*?stand-alone \(([\w-]+)\)", text, re.M):
        H, W, pu, N, B = (int(m.group(k)) for k in range(1, 6))
        row = (H, W, pu, N, B, m.group(6), m.group(7))
        if row not in out:                                    # (a table may hold a shape in several sections: with / without the profiler)
            out.append(row)
    return files[-1], out


def cases(lib):
    name, shapes = profile_shapes()
    out = []
    for H, W, pu, N, B, form, alone in shapes:
        out.append(dict(source=f"profiles/{name}", H=H, W=W, pu=pu, N=N, B=B, Hc=256, recorded_form=form, recorded_form_alone=alone, plan=plan(lib, H, W, pu, N, B)))
    extra = [
        ("kSideStreamMinBytes, one launch for fill + drawing at every batch size", [dict(H=8, W=8, pu=32, N=256, B=b) for b in (1, 16, 64, 256, 1024, 4096)]),
        ("kSideStreamMinBytes, another camera height: the side stream pays from 256 MiB of top view", [dict(H=8, W=8, pu=32, N=256, B=b, Hc=128) for b in (16, 255, 256, 1024, 4096)]),
        ("kPartsMax / kPartsMinRays, kDrawWideBlockLds: few big images", [dict(H=32, W=32, pu=32, N=1024, B=b) for b in (16, 64, 128, 192, 256, 512)]
         + [dict(H=24, W=24, pu=32, N=256, B=b) for b in (57, 114, 228, 341, 455, 910)]),
        ("kAloneBlock64Agents / kAloneBlock128Agents: the stand-alone draw kernel's block", [dict(H=8, W=8, pu=10, N=256, B=b) for b in (4096, 12287, 12288, 24575, 24576, 41943)]
         + [dict(H=8, W=8, pu=16, N=256, B=16384)]),
        ("kRuns*: several GiB of top view whose lines are long against the camera view", [dict(H=16, W=16, pu=32, N=512, B=b) for b in (1024, 2048, 4096, 16384)]
         + [dict(H=32, W=32, pu=32, N=1024, B=b) for b in (1024, 8192)]),
        ("kRingThreeBuffersLds / kRingLdsCap / kLineWalkMaxPixels: the ring and the in-place form", [dict(H=8, W=8, pu=p, N=64, B=64) for p in (4, 5, 7, 40, 72, 100, 136, 160, 200)]
         + [dict(H=64, W=4, pu=300, N=64, B=4)]),
        ("kDrawFirst (kFill*, kDraw*): the fill is the shorter of the two", [dict(H=8, W=8, pu=32, N=256, B=4096, Hc=128), dict(H=16, W=16, pu=32, N=256, B=1024, Hc=128),
                                                                          dict(H=8, W=8, pu=32, N=256, B=4096, Hc=300), dict(H=8, W=16, pu=32, N=256, B=2048, Hc=128)]),
    ]
    for why, lst in extra:
        for c in lst:
            c = dict(Hc=256, **c) if "Hc" not in c else c
            out.append(dict(source=why, **c, plan=plan(lib, c["H"], c["W"], c["pu"], c["N"], c["B"], c["Hc"])))
    return out


def main():
    lib = _capi.load("dev")
    data = dict(hw=HW, note="written by tools/make_top_view_plan_cases.py from the development build's rcw_dev_plan_top_view; a deliberate retune rewrites it",
                cases=cases(lib))
    path = os.path.join(ROOT, "tests", "golden", "top_view_plan_cases.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print(f"{path}: {len(data['cases'])} cases")
    bad = [c for c in data["cases"] if "recorded_form" in c and (c["recorded_form"], c["recorded_form_alone"]) != (c["plan"]["form"], c["plan"]["form_alone"])]
    for c in bad:
        print("  DIFFERS from the profile table:", {k: c[k] for k in ("H", "W", "pu", "N", "B")}, c["recorded_form"], c["recorded_form_alone"], "->", c["plan"]["form"], c["plan"]["form_alone"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
