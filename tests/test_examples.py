"""The examples and the README's snippet run as a user would run them (GPU box): each in a process of its own."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, **kw):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=300, **kw)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_random_policy_example(tmp_path):
    out = _run(["examples/random_policy.py", "96", "40"])
    assert "('MOVE_FORWARD', 'MOVE_BACKWARD', 'TURN_LEFT', 'TURN_RIGHT')" in out or "MOVE_FORWARD" in out
    assert re.search(r"96 agents x 40 random steps on .*gfx950|96 agents x 40 random steps", out), out
    assert os.path.getsize("/tmp/agent0.ppm") > 256 * 512 * 3          # (reference defaults: 512 columns x 256 rows, binary PPM)


def test_torch_loop_example():
    out = _run(["examples/torch_loop.py", "512", "30"])
    assert re.search(r"512 agents x 30 steps, policy \+ engine on one stream: [\d.]+ M env-steps/s", out), out


def test_readme_snippet(tmp_path):
    """The python block of README.md, with `actions` defined, runs as printed."""
    text = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    script = tmp_path / "readme_snippet.py"
    script.write_text("import sys, numpy as np\nsys.path.insert(0, %r)\nactions = np.random.default_rng(0).integers(1, 5, 4096).astype(np.uint8)\n" % ROOT
                      + block + "\nprint('state', obs.shape, RCW.RLBase.reward(rl).shape, RCW.RLBase.is_terminated(rl).shape)\nenv.close()\n")
    out = _run([str(script)])
    assert "state (4096, 256, 256) (4096,) (4096,)" in out, out


@pytest.mark.parametrize("flags", [["--workload", "cfg2"], ["--workload", "cfg3"], ["--workload", "cfg4"], ["--workload", "cfg5"], ["--top-view"],
                                   ["--api", "rlbase"], ["--batch", "1000"], ["--no-auto-reset"], ["--gather"]])
def test_bench_modes(flags):
    """Every mode of bench.py the documents quote runs and prints ONE JSON line with the contract's keys (a few steps each; the
    default invocation is what the round-end driver runs)."""
    import json

    out = _run(["bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--traffic", "off"] + flags)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, (k, flags)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "env-steps/s" and d["dtype"] == "f32"
    assert 0.3 < d["roofline"]["frac"] < 1.0, d["roofline"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["ms_per_step_min"] == d["ms_per_step_max"] == d["ms_per_step"] and len(d["roofline"]["per_rank"]) == 1   # one rank: no spread
    if flags == ["--top-view"]:
        # the block names the kernel that RAN between the two events (camera fill + drawing in one launch),
        # with that launch's bytes (frames + the drawing's scratch) — never the plain fill's label or its committed traffic figure
        r = d["roofline"]
        assert r["kernel"] == "rcw_fill256_draw_kernel" and "launch_note" in r
        assert r["bytes_per_launch"] == 4 * 256 * 256 * 4096 + 4096 * (8192 + 8 + 64)
        assert r["traffic"] is None and r["traffic_source"] is None          # (--traffic off here; live it is this kernel's own)
        assert d["top_view"]["form"] == "two-kernels" and 0.3 < d["top_view"]["frac"] < 1.0
    else:
        assert d["roofline"]["kernel"] != "rcw_fill256_draw_kernel" and "launch_note" not in d["roofline"]
    if flags == ["--gather"]:
        assert "gather" in d and "error" not in d["gather"], d.get("gather")
