#!/usr/bin/env python3
"""Dev tool: one short line from bench.py's JSON line on stdin (step and fill kernel times in µs).

    python bench.py --top-view --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py <label>
"""
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "step", round(d["ms_per_step"]*1e3,1), "fill", round(d["roofline"]["launch_ms"]*1e3,1))
