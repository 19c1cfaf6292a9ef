"""End-to-end time of the drop-in Module (TriRenderer / TetRenderer: casts, transposes, inverses, autograd) against the bare
_C calls, in the three host modes: default (one host wait per call for the size read-back), asynchronous
(_C.set_async: no host wait, capacity from the previous call) and as ONE captured HIP graph (forward + backward
through autograd inside torch.cuda.graph; replay = one launch).  usage: python scripts/time_module.py [C1|C2|C3|C4] [json-out]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch as th
import dmesh_renderer_amd as dmr
from dmesh_renderer_amd import _C, scenes
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
tet = cfg.kind == "tet"
B, H, W = cfg.B, cfg.H, cfg.W
t = {k: v.to(dev) for k, v in d.items()}
gc, gd = scenes.upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
names = ("verts_color", "faces_opacity") if tet else ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")
leaves = {k: t[k].clone().requires_grad_(True) for k in names}
faces = t["faces"].to(th.int32)
if tet:
    r = dmr.TetRenderer(dmr.TetRenderSettings(H, W, t["bg"], 0))
    topo = [t[k].to(th.int32) for k in ("tets", "face_tets", "tet_faces")]
    def render():
        return r(t["verts"], faces, leaves["verts_color"], leaves["faces_opacity"], t["mv_mats"], t["proj_mats"], t["verts_depth"],
                 t["faces_intense"], *topo)[:2]
else:
    r = dmr.TriRenderer(dmr.TriRenderSettings(H, W, t["bg"]))
    def render():
        return r(leaves["verts"], faces, leaves["verts_color"], leaves["faces_opacity"], t["mv_mats"], t["proj_mats"],
                 leaves["verts_depth"], leaves["faces_intense"])
def module_step():
    for v in leaves.values(): v.grad = None
    color, depth = render()
    th.autograd.backward([color, depth], [gc, gd])
args = scenes.c_args(d, dev, tet=tet)
def c_step():
    if tet:
        o = _C.render_tets(*args, H, W, 0); _C.render_tets_backward(*args, gc, gd, *o[3:7])
    else:
        o = _C.render_tris(*args, H, W); _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7])
def inv_only():
    th.inverse(t["mv_mats"].transpose(1, 2)); th.inverse(t["proj_mats"].transpose(1, 2))
def timeit(fn, n=50):
    for _ in range(8): fn()
    th.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
res = {"config": cfgname}
res["_C fwd+bwd"] = timeit(c_step)
res["Module fwd+bwd (autograd)"] = timeit(module_step)
res["2 x th.inverse"] = timeit(inv_only)
_C.set_async(True)
res["_C fwd+bwd, async"] = timeit(c_step)
res["Module fwd+bwd, async"] = timeit(module_step)
_C.set_async(False)
assert not _C.overflowed()
# the whole step as one HIP graph: warm up on a side stream (the default calls there also provide the size estimates
# the captured, never-waiting calls need), capture, replay
s = th.cuda.Stream()
s.wait_stream(th.cuda.current_stream())
with th.cuda.stream(s):
    for _ in range(3): module_step()
th.cuda.current_stream().wait_stream(s)
for v in leaves.values(): v.grad = None
g = th.cuda.CUDAGraph()
with th.cuda.graph(g):
    color, depth = render()
    th.autograd.backward([color, depth], [gc, gd])
res["Module fwd+bwd, graph replay"] = timeit(g.replay)
assert not _C.overflowed()
for k, v in res.items():
    print(f"{k:34s} {v if isinstance(v, str) else round(v, 4)}" + ("" if isinstance(v, str) else " ms"))
if len(sys.argv) > 2:
    json.dump({k: (v if isinstance(v, str) else round(v, 4)) for k, v in res.items()}, open(sys.argv[2], "w"), indent=1)
