"""Per (view, tile row) statistics of one full forward: list entries, blended (pixel, face) pairs, busy tiles -- what
sharding's cost model is fitted on.  usage: python scripts/dump_row_stats.py C5 out.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch as th
from dmesh_renderer_amd import _C, scenes
cfgname, out = sys.argv[1], sys.argv[2]
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
gy, gx = (H + 15) // 16, (W + 15) // 16
args = scenes.c_args(d, dev)
o = _C.render_tris(*args, H, W)
rng = _C.export("ranges", args, False, o[0], o[3:7], H, W, th.int32).cpu().numpy().reshape(B, gy, gx, 2).astype(np.int64)
hits = _C.export("tile_hits", args, False, o[0], o[3:7], H, W, th.int32).cpu().numpy().reshape(B, gy, gx).astype(np.int64)
lens = rng[..., 1] - rng[..., 0]
np.savez(out, entries=lens.sum(2), pairs=hits.sum(2), busy=(lens > 0).sum(2), chunks=((lens + 127) // 128).sum(2), gx=gx, F=d["faces"].shape[0])
print(cfgname, "entries", lens.sum(), "pairs", hits.sum(), "busy", (lens > 0).sum())
