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
for partition in ("bands", "view_bands"):   # bands of both views per rank / (view, band) segments (B = 2 views, 2 ranks)
    for assemble in (True, False):
        sh = sharding.ShardedTriRenderer(settings, assemble=assemble, partition=partition)
        assert sh.world == world == 2 and sh.rows != (0, 0)
        if partition == "view_bands":  # the second view costs twice the first: rank 0 renders view 0 and the top of view 1
            gy = sharding.tile_rows(H)
            sh.set_row_work(np.stack([np.ones(gy), 2.0 * np.ones(gy)]))
            sh.segment_cost_per_face = 0.0
            parts = sh.view_parts(B, t["faces"].shape[0])
            assert [len(p) for p in parts] == [2, 1], parts
        c, z, g = run(sh)
        if assemble:
            assert th.equal(c, full[0]) and th.equal(z, full[1]), "assembled image differs"
        elif partition == "bands":
            r0, r1 = sh.rows
            assert th.equal(c[:, :, 16 * r0:16 * r1], full[0][:, :, 16 * r0:16 * r1]), "band rows differ"
        else:
            for v, r0, r1 in parts[rank]:
                assert th.equal(c[v, :, 16 * r0:16 * r1], full[0][v, :, 16 * r0:16 * r1]), "this rank's segment differs"
        for a, b, k in zip(g, full[2], names):
            e = scenes.rel_err(a.cpu().numpy(), b.cpu().numpy())
            assert e <= scenes.sum_order_tol(k), (partition, k, e)

# the tet renderer, same sharding (ShardedTetRenderer: bands, one all-gather of the images, ONE all-reduce over [3P | F])
Ht = Wt = 160
dt = scenes.kuhn_tets(5, B, Ht, Wt, seed=3)
tt = {k: v.to(dev) for k, v in dt.items()}
gct, gdt = scenes.upstream_grads(B, Ht, Wt)
gct, gdt = gct.to(dev), gdt.to(dev)
tsettings = dmr.TetRenderSettings(Ht, Wt, tt["bg"], 0)


def run_tet(renderer):
    vc = tt["verts_color"].clone().requires_grad_(True); fo = tt["faces_opacity"].clone().requires_grad_(True)
    color, depth, active = renderer(tt["verts"], tt["faces"], vc, fo, tt["mv_mats"], tt["proj_mats"], tt["verts_depth"],
                                    tt["faces_intense"], tt["tets"], tt["face_tets"], tt["tet_faces"])
    th.autograd.backward([color, depth], [gct, gdt])
    return color.detach(), depth.detach(), active, [vc.grad, fo.grad]


tfull = run_tet(dmr.TetRenderer(tsettings))
assert bool(tfull[2].any())
sht = sharding.ShardedTetRenderer(tsettings, assemble=True)
assert sht.world == 2 and sht.rows != (0, 0)
c, z, a, g = run_tet(sht)
assert th.equal(c, tfull[0]) and th.equal(z, tfull[1]) and th.equal(a, tfull[2]), "assembled tet image differs"
for x, y, k in zip(g, tfull[3], ("verts_color", "faces_opacity")):
    e = scenes.rel_err(x.cpu().numpy(), y.cpu().numpy())
    assert e <= 1e-5, (k, e)
dist.barrier()
if rank == 0:
    print("sharded ok")
dist.destroy_process_group()
