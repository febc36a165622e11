import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycastworlds_jl_amd as RCW
HEIGHTS = ((256, 4096), (128, 8192), (512, 2048), (64, 16384), (100, 10486), (1024, 1024), (84, 12483), (250, 4194), (300, 3495), (40, 26214), (480, 2184), (720, 1456))
if len(sys.argv) > 1:                                    # chosen heights: H_cam,B H_cam,B ...
    HEIGHTS = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for hc, B in HEIGHTS:
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256, height_camera_view_pu=hc)
    st = torch.cuda.Stream(); env.set_stream(st.cuda_stream); torch.cuda.set_stream(st)
    a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
    for _ in range(5): RCW.act_(env, a)
    env.profile(True)
    for _ in range(int(os.environ.get("HCAM_STEPS", "50"))): RCW.act_(env, a)
    c, t, f, n = env.profile_read(); env.profile(False)
    by = 4 * hc * 256 * B
    print(f"H_cam {hc:5d} B {B:6d} {env.fill_kernel_name():24s}: fill {f*1e3:7.1f} us  {by / f / 1e6:7.0f} GB/s  ({by / f / 1e6 / 80:.1f} % of 8 TB/s)", flush=True)
    env.sync(); env.close()
