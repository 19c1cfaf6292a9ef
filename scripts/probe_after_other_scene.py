"""Probe: does a different scene with the same view configuration leave the C4 step slower?  (default calls, per-stage events)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch as th
from dmesh_renderer_amd import _C, scenes
from dmesh_renderer_amd.scenes import c_args, upstream_grads
dev = th.device("cuda:0")
cfg = scenes.CONFIGS["C4"]; B, H, W = cfg.B, cfg.H, cfg.W
gc, gd = upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
def mk(**kw):
    return c_args(scenes.make("C4", **kw), dev)
A = mk(); E = mk(opacity=(0.5, 0.95))
def step(args):
    o = _C.render_tris(*args, H, W)
    return o, _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7])
def measure(args, tag, n=40):
    for _ in range(5): step(args)
    th.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(args)
    th.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    _C.profile_enable(0xFFFFFFFF)
    for _ in range(10): step(args)
    th.cuda.synchronize(); _C.profile_enable(0)
    ms, cnt = _C.profile_collect()
    st = {_C.stage_name(i): round(ms[i] / cnt[i] * 1e3, 1) for i in range(_C.NUM_STAGES) if cnt[i]}
    print(tag, round(dt * 1e3, 4), "ms/step; redo", _C.redo_count(), st, flush=True)
for _ in range(200): step(A)
measure(A, "C4 first      ")
measure(E, "early-out     ")
measure(A, "C4 again      ")
del E; th.cuda.empty_cache()
measure(A, "C4, cache emptied")
