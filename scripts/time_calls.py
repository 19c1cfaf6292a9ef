"""Wall time of the two _C calls (with a device sync after each) vs the sum of the library's stage timers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch as th
from dmesh_renderer_amd import _C, scenes
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
args = scenes.c_args(d, dev); gc, gd = scenes.upstream_grads(B, H, W); gc, gd = gc.to(dev), gd.to(dev)
for it in range(6):
    _C.profile_enable(0xFFFFFFFF)
    th.cuda.synchronize(); t0 = time.perf_counter()
    o = _C.render_tris(*args, H, W); t1 = time.perf_counter()
    th.cuda.synchronize(); t2 = time.perf_counter()
    g = _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7]); t3 = time.perf_counter()
    th.cuda.synchronize(); t4 = time.perf_counter()
    _C.profile_enable(0)
    ms, cnt = _C.profile_collect()
    print(f"it{it}: fwd call {1e3*(t1-t0):.3f} ms (+sync {1e3*(t2-t1):.3f}), bwd call {1e3*(t3-t2):.3f} ms (+sync {1e3*(t4-t3):.3f}), kernels {sum(ms):.3f} ms, "
          f"mem alloc {th.cuda.memory_allocated()/1e9:.2f} GB reserved {th.cuda.memory_reserved()/1e9:.2f} GB")
    del o, g
