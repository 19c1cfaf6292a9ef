"""(Belongs to profiles/r03/fused_records_experiment.patch; on this tree all three calls take the same path.)  The forward-record
path (second and later calls of a view configuration) against the round-2 path (first call) and the
oracle, with the per-stage times of both.   python tests/tools/rec_check.py [C1 C2 C4 early ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch as th
from dmesh_renderer_amd import scenes, _C
from oracle import oracle as O

NAMES = ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")


def run(name):
    over = {}
    cfgname = name
    if name == "early": cfgname, over = "C2", {"opacity": (0.5, 0.95)}
    if name == "high": cfgname, over = "C2", {"opacity": (0.9, 0.9999)}
    if name == "uni": cfgname, over = "C2", {"opacity": (0.0, 1.0)}
    cfg = scenes.CONFIGS[cfgname]
    d = scenes.make(cfgname, **over)
    dev = th.device("cuda:0")
    args = scenes.c_args(d, dev)
    gc, gd = scenes.upstream_grads(cfg.B, cfg.H, cfg.W)
    gc, gd = gc.to(dev), gd.to(dev)
    O.build(); O.lib(); oracle = O
    sc = oracle.scene_from_module_inputs(d, cfg.H, cfg.W)
    ocolor, odepth, ost = oracle.tri_forward(sc)
    og = oracle.tri_backward(sc, ost, gc.cpu().numpy(), gd.cpu().numpy())
    res = []
    for call in range(3):
        r0 = _C.redo_count() if hasattr(_C, "redo_count") else 0
        out = _C.render_tris(*args, cfg.H, cfg.W)
        g = _C.render_tris_backward(*args, gc, gd, out[0], *out[3:7])
        th.cuda.synchronize()
        ferr = float(np.abs(out[1].cpu().numpy() - ocolor).max())
        errs = {k: scenes.rel_err(got.cpu().numpy(), og[k]) for got, k in zip(g, NAMES)}
        close = {k: bool(scenes.elementwise_close(got.cpu().numpy(), og[k])) for got, k in zip(g, NAMES)}
        res.append([x.clone() for x in g])
        print(f"{name} call {call}: R {out[0]} fwd err {ferr:.2e} grad err " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()) +
              f" per-entry ok {all(close.values())} {[k for k, v in close.items() if not v]}", flush=True)
    for k, a, b_ in zip(NAMES, res[0], res[2]):
        den = max(1.0, float(a.abs().max()))
        print(f"   {k}: |records - round-2 path| / max = {float((a - b_).abs().max()) / den:.2e}")
    # stage times
    _C.profile_enable(0xfff)
    for _ in range(10):
        out = _C.render_tris(*args, cfg.H, cfg.W)
        g = _C.render_tris_backward(*args, gc, gd, out[0], *out[3:7])
    th.cuda.synchronize()
    ms, n = _C.profile_collect()
    _C.profile_enable(0)
    print("   stages (us): " + "  ".join(f"{_C.stage_name(i)} {1e3 * m / max(c, 1):.1f}" for i, (m, c) in enumerate(zip(ms, n)) if c))
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        out = _C.render_tris(*args, cfg.H, cfg.W)
        g = _C.render_tris_backward(*args, gc, gd, out[0], *out[3:7])
    th.cuda.synchronize()
    print(f"   {1e3 * (time.perf_counter() - t0) / K:.4f} ms per fwd+bwd (default calls)", flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["C1", "C2", "C4"]):
        run(n)
