import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "step", round(d["ms_per_step"]*1e3,1), "top", round(d["top_view"]["launch_ms"]*1e3,1), "fill", round(d["roofline"]["launch_ms"]*1e3,1))
