"""Per-stage timing of the tet renderer on a Kuhn-lattice scene (default C3: m=16, 800x800)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch as th
from dmesh_renderer_amd import _C, scenes

ap = argparse.ArgumentParser(); ap.add_argument("--m", type=int, default=16); ap.add_argument("--size", type=int, default=800)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = th.device("cuda:0"); H = W = a.size
d = scenes.kuhn_tets(a.m, 1, H, W)
args = scenes.c_args(d, dev, tet=True)
gc, gd = scenes.upstream_grads(1, H, W); gc, gd = gc.to(dev), gd.to(dev)
def step():
    o = _C.render_tets(*args, H, W, 0)
    g = _C.render_tets_backward(*args, gc, gd, *o[3:7])
    return o, g
for _ in range(3): step()
_C.profile_enable(0xFFFFFFFF); th.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): o, g = step()
th.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
_C.profile_enable(0)
ms, cnt = _C.profile_collect()
st = {_C.stage_name(i): round(ms[i] / cnt[i], 4) for i in range(_C.NUM_STAGES) if cnt[i]}
print(json.dumps({"tets": int(d["tets"].shape[0]), "faces": int(d["faces"].shape[0]), "image": [H, W], "ms_per_step": round(dt * 1e3, 4),
                  "Mpix_s": round(H * W / dt / 1e6, 1), "active_frac": round(float(o[2].mean()), 3), "stages_ms": st}))
