"""cProfile of the host side of the Module path at a small size (where Python, not the GPU, sets the step time)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch as th
import dmesh_renderer_amd as dmr
from dmesh_renderer_amd import scenes
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C1"
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
t = {k: v.to(dev) for k, v in d.items()}
gc, gd = scenes.upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
leaves = {k: t[k].clone().requires_grad_(True) for k in ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")}
r = dmr.TriRenderer(dmr.TriRenderSettings(H, W, t["bg"]))
faces = t["faces"].to(th.int32)
def step():
    for v in leaves.values(): v.grad = None
    color, depth = r(leaves["verts"], faces, leaves["verts_color"], leaves["faces_opacity"], t["mv_mats"], t["proj_mats"],
                     leaves["verts_depth"], leaves["faces_intense"])
    th.autograd.backward([color, depth], [gc, gd])
for _ in range(20): step()
th.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
th.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
