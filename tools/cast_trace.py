"""Dev tool (GPU box): where a wavefront of rcw_cast_kernel spends its life.  Needs the measurement build
(`make -C raycastworlds.jl_amd/csrc trace` -> lib/librcw_hip_trace.so, loaded here by path; the package never loads it): the first
wavefront of every workgroup (= agent) leaves s_memrealtime (100 MHz, one clock for the device) at entry, when its state (load
batch 1) is back, behind the barrier, behind the dynamics, when the ray-table row (batch 2) is back, behind its first column,
behind its last, and when its stores are acknowledged.

    python tools/cast_trace.py [workload=cfg2] [reps=8]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raycastworlds_jl_amd as RCW
from bench import WORKLOADS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RCW_LIBRARY") or os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "librcw_hip_trace.so")
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kw, B = WORKLOADS[workload]
env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, library=LIB, **kw)
lib = env._lib
lib.rcw_cast_trace_read.argtypes = [ctypes.c_void_p]
acts = torch.randint(1, 5, (64, B), dtype=torch.uint8, device="cuda")
buf = np.zeros(4096 * 10, dtype=np.uint64)
n = min(B, 4096)
names = ["state back (batch 1)", "tiles staged, barrier", "dynamics", "table row back (batch 2)", "first column", "other columns", "stores acknowledged"]
print(f"rcw_cast_kernel at {workload}: {B} agents, {kw}; first wavefront of the first {n} workgroups; us")
print(" launch | first entry -> last end | entries spread | wave life median / p95 | " + " | ".join(names))
rows = []
for rep in range(reps):
    for s in range(6):
        RCW.act_(env, acts[(6 * rep + s) % 64])
    try:
        env.sync()
    except IndexError:
        env.clear_error()
    assert lib.rcw_cast_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(4096, 10)[:n].astype(np.int64)
    st = t[:, :8]
    t0 = st[:, 0].min()
    life = (st[:, 7] - st[:, 0]) / 100.0
    seg = np.diff(st, axis=1) / 100.0                     # 7 segments
    span = (st[:, 7].max() - t0) / 100.0
    entries = (st[:, 0].max() - t0) / 100.0
    rows.append(np.concatenate([[span, entries, np.median(life), np.percentile(life, 95)], np.median(seg, axis=0)]))
    print(f" {rep:6d} | {span:23.2f} | {entries:14.2f} | {np.median(life):9.2f} / {np.percentile(life, 95):5.2f}   | " +
          " | ".join(f"{np.median(seg[:, k]):{len(names[k])}.2f}" for k in range(7)))
r = np.median(np.array(rows[1:]), axis=0)
print("median of launches 1.. : span %.2f, entries spread %.2f, life %.2f (p95 %.2f); segments " % tuple(r[:4]) + " ".join(f"{v:.2f}" for v in r[4:]))
xcc = (t[:, 8] >> 32) & 0xF
print("last launch: entry time by XCD (median us after the first entry): " + " ".join(f"{int(x)}:{np.median((st[xcc == x, 0] - t0) / 100.0):.2f}" for x in sorted(set(xcc.tolist()))))
print("last launch: workgroups entering within 0.5 / 1 / 2 / 4 us: " + " / ".join(str(int(((st[:, 0] - t0) / 100.0 <= v).sum())) for v in (0.5, 1, 2, 4)))
try:
    env.sync()
except IndexError:
    env.clear_error()
env.close()
