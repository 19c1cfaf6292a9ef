"""Where the drop-in Module's time goes beyond the two `_C` calls (C1 / C2 sizes: the Module is host-bound there).
usage: python scripts/time_module_parts.py [C1|C2|C4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch as th
import dmesh_renderer_amd as dmr
from dmesh_renderer_amd import _C, scenes
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C1"
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
t = {k: v.to(dev) for k, v in d.items()}
gc, gd = scenes.upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
names = ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")
leaves = {k: t[k].clone().requires_grad_(True) for k in names}
faces = t["faces"].to(th.int32)
r = dmr.TriRenderer(dmr.TriRenderSettings(H, W, t["bg"]))
def render():
    return r(leaves["verts"], faces, leaves["verts_color"], leaves["faces_opacity"], t["mv_mats"], t["proj_mats"],
             leaves["verts_depth"], leaves["faces_intense"])
def module_step():
    for v in leaves.values(): v.grad = None
    color, depth = render()
    th.autograd.backward([color, depth], [gc, gd])
def fwd_nograd():
    with th.no_grad():
        render()
def fwd_grad():
    render()
args = scenes.c_args(d, dev)
def c_step():
    o = _C.render_tris(*args, H, W); _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7])
def c_fwd():
    _C.render_tris(*args, H, W)
mvt, prt = t["mv_mats"].transpose(1, 2), t["proj_mats"].transpose(1, 2)
def inv():
    _C.invert_mats(mvt, prt)
class Noop(th.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, d_, e):
        ctx.save_for_backward(a, b, c, d_, e)
        return a.view_as(a), b.view_as(b)
    @staticmethod
    def backward(ctx, ga, gb):
        s = ctx.saved_tensors
        return ga, gb, None, None, None
x = [th.zeros(4, device=dev, requires_grad=True) for _ in range(5)]
gx = th.ones(4, device=dev)
def noop_step():
    a, b = Noop.apply(*x)
    th.autograd.backward([a, b], [gx, gx])
def timeit(fn, n=200):
    for _ in range(20): fn()
    th.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for name, fn in (("_C fwd", c_fwd), ("_C fwd+bwd", c_step), ("invert_mats", inv), ("Module fwd, no_grad", fwd_nograd),
                 ("Module fwd, grad mode", fwd_grad), ("Module fwd+bwd", module_step), ("no-op autograd.Function fwd+bwd", noop_step)):
    print(f"{cfgname} {name:36s} {timeit(fn):.4f} ms", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "profile":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(50): module_step()
        th.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=30, max_name_column_width=60))
