import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycastworlds_jl_amd as RCW
"""Dev tool (GPU box): the camera fill at other camera heights (~1 GiB of frames a step, 256 view columns): the fill kernel's launch by HIP events and the
WHOLE step (events around all steps) — in both forms of the step where the height takes the one-launch form (256 k, 128, 64 rows).
    python tools/hcam_bench.py [H_cam,B ...]"""
HEIGHTS = ((256, 4096), (128, 8192), (512, 2048), (64, 16384), (100, 10486), (1024, 1024), (84, 12483), (250, 4194), (300, 3495), (40, 26214), (480, 2184), (720, 1456))
if len(sys.argv) > 1:                                    # chosen heights: H_cam,B H_cam,B ...
    HEIGHTS = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
STEPS = int(os.environ.get("HCAM_STEPS", "50"))
for hc, B in HEIGHTS:
    out = []
    for form in ("rule", "two-launches"):
        env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256, height_camera_view_pu=hc)
        if form == "two-launches":
            if out and out[0][0] == "two-launches":
                env.close(); continue                   # (the rule already took it)
            env.set_step_form("two-launches")
        st = torch.cuda.Stream(); env.set_stream(st.cuda_stream); torch.cuda.set_stream(st)
        a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
        for _ in range(5): RCW.act_(env, a)
        env.profile(True)
        for _ in range(STEPS): RCW.act_(env, a)
        c, t, f, n = env.profile_read(); env.profile(False)
        env.sync(); env.timer_start()
        for _ in range(STEPS): RCW.act_(env, a)
        step = env.timer_stop() / STEPS
        out.append((env.step_form(), env.fill_kernel_name(), f, step))
        env.sync(); env.close()
    by = 4 * hc * 256 * B
    s = f"H_cam {hc:5d} B {B:6d} "
    for sf, name, f, step in out:
        s += f"| {sf:12s} {name:24s} launch {f*1e3:7.1f} us ({by / f / 1e6 / 80:4.1f} %), whole step {step*1e3:7.1f} us ({by / step / 1e6 / 80:4.1f} %) "
    print(s, flush=True)
