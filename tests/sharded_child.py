"""Child of tests/test_sharded_gpu.py: one rank of a 2-rank (gloo) run of ShardedTriRenderer on one GPU.
Every rank also renders the full image alone and compares."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch as th
import torch.distributed as dist
import dmesh_renderer_amd as dmr
from dmesh_renderer_amd import scenes, sharding

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = th.device("cuda:0")
th.cuda.set_device(dev)
B, H, W = 2, 200, 328
d = scenes.layered_sheets(3, 14, B, H, W, seed=9)
t = {k: v.to(dev) for k, v in d.items()}
gc, gd = scenes.upstream_grads(B, H, W)
gc, gd = gc.to(dev), gd.to(dev)
names = ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")
settings = dmr.TriRenderSettings(H, W, t["bg"])


def run(renderer):
    leaves = {k: t[k].clone().requires_grad_(True) for k in names}
    color, depth = renderer(leaves["verts"], t["faces"], leaves["verts_color"], leaves["faces_opacity"], t["mv_mats"],
                            t["proj_mats"], leaves["verts_depth"], leaves["faces_intense"])
    th.autograd.backward([color, depth], [gc, gd])
    return color.detach(), depth.detach(), [leaves[k].grad for k in names]


full = run(dmr.TriRenderer(settings))
for assemble in (True, False):
    sh = sharding.ShardedTriRenderer(settings, assemble=assemble)
    assert sh.world == world == 2 and sh.rows != (0, 0)
    c, z, g = run(sh)
    if assemble:
        assert th.equal(c, full[0]) and th.equal(z, full[1]), "assembled image differs"
    else:
        r0, r1 = sh.rows
        assert th.equal(c[:, :, 16 * r0:16 * r1], full[0][:, :, 16 * r0:16 * r1]), "band rows differ"
    for a, b, k in zip(g, full[2], names):
        e = scenes.rel_err(a.cpu().numpy(), b.cpu().numpy())
        assert e <= 1e-5, (k, e)
dist.barrier()
if rank == 0:
    print("sharded ok")
dist.destroy_process_group()
