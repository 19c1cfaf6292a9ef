"""Stage times with a few screen-filling triangles added to C2 (binning cliff: one thread emitting a whole rect)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch as th
from dmesh_renderer_amd import _C, scenes
nbig = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = scenes.CONFIGS["C2"]; d = scenes.make("C2"); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
big = scenes.layered_sheets(nbig // 2, 2, B, H, W, seed=5, opacity=(0.01, 0.03))   # nbig triangles, one quad per layer
big["verts"] = big["verts"] * th.tensor([6.0, 6.0, 1.0])
P0 = d["verts"].shape[0]
m = dict(d)
m["verts"] = th.cat([d["verts"], big["verts"]]); m["verts_color"] = th.cat([d["verts_color"], big["verts_color"]])
m["faces"] = th.cat([d["faces"], big["faces"] + P0]); m["faces_opacity"] = th.cat([d["faces_opacity"], big["faces_opacity"]])
m["verts_depth"] = th.cat([d["verts_depth"], big["verts_depth"]], dim=1); m["faces_intense"] = th.cat([d["faces_intense"], big["faces_intense"]], dim=1)
for name, sc in (("C2", d), (f"C2 + {nbig} screen-filling triangles", m)):
    args = scenes.c_args(sc, dev); gc, gd = scenes.upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
    def step():
        o = _C.render_tris(*args, H, W); _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7]); return o
    for _ in range(3): o = step()
    _C.profile_enable(0xFFFFFFFF); th.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    th.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    _C.profile_enable(0)
    ms, cnt = _C.profile_collect()
    st = {_C.stage_name(i): round(ms[i] / cnt[i], 4) for i in range(_C.NUM_STAGES) if cnt[i]}
    print(name, "R", o[0], "ms/step", round(dt * 1e3, 3), json.dumps(st))
