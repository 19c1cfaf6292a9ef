"""Test infrastructure. One-off long fuzz run (not collected by pytest): random triangle soups and Delaunay tets at random image
sizes / view counts / row bands against the oracle, for a time budget.  Prints one line per case; exits non-zero on
the first mismatch with the parameters needed to reproduce it.
    python tests/tools/fuzz_campaign.py --seconds 240 --seed0 1000"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch as th
from dmesh_renderer_amd import _C
from util import c_args, rel_err, upstream_grads
from oracle import oracle as O
from test_fuzz_gpu import _soup, _delaunay, big_case, NAMES
from util import elementwise_close

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=240); ap.add_argument("--seed0", type=int, default=1000)
ap.add_argument("--big", action="store_true", help="fewer, larger tri cases: layered sheets up to ~400k faces, images up to 1600^2 (more than 8192 tiles), several views, random tile-row bands")
a = ap.parse_args()
O.build(); O.lib(); oracle = O; dev = th.device("cuda:0")
t0 = time.time(); seed = a.seed0; ncase = 0; n_marginal = 0
while time.time() - t0 < a.seconds:
    rng = np.random.RandomState(seed)
    B = int(rng.randint(1, 4)); H = int(rng.randint(17, 260)); W = int(rng.randint(17, 300))
    tet = rng.rand() < 0.3 and not a.big
    rows = (0, 0)
    if a.big:
        d, B, H, W, rows = big_case(seed)
    gc, gd = upstream_grads(B, H, W)
    if not tet:
        if a.big:
            P, F = d["verts"].shape[0], d["faces"].shape[0]
        else:
            P = int(rng.randint(8, 600)); F = int(rng.randint(30, 3000))
            d = _soup(seed, P, F, B, H, W)
            if rng.rand() < 0.3: d["verts"] = d["verts"] * float(rng.uniform(0.05, 4.0))
        sc = oracle.scene_from_module_inputs(d, H, W, rows=rows)
        ocolor, odepth, ost = oracle.tri_forward(sc)
        args = c_args(d, dev)
        out = _C.render_tris(*args, H, W, rows=rows); th.cuda.synchronize()
        R, bufs = out[0], out[3:7]
        ex = lambda n, dt: _C.export(n, args, False, R, bufs, H, W, dt).cpu().numpy()
        nc_g, nc_o = ex("n_contrib", th.int32).view(np.uint32).reshape(B, H, W), ost.get("n_contrib").reshape(B, H, W)
        if rows != (0, 0):  # only the band's pixel rows are defined
            y0, y1 = rows[0] * 16, min(H, rows[1] * 16)
            nc_g, nc_o = nc_g[:, y0:y1], nc_o[:, y0:y1]
        checks = {"R": R == ost.num_rendered,
                  "face_list": np.array_equal(ex("face_list", th.int32).view(np.uint32), ost.get("values")),
                  "n_contrib": np.array_equal(nc_g, nc_o)}
        ok = all(checks.values())
        if not ok: print("   mismatch:", checks, "oracle R", ost.num_rendered, flush=True)
        fin = np.isfinite(ocolor).all(axis=1, keepdims=True)
        ferr = float(np.abs(np.where(fin, out[1].cpu().numpy() - ocolor, 0.0)).max())
        og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
        g = _C.render_tris_backward(*args, gc.to(dev), gd.to(dev), R, *bufs, rows=rows); th.cuda.synchronize()
        gerr = 0.0
        for got, k in zip(g, NAMES):
            x, r = got.cpu().numpy(), og[k]; f = np.isfinite(r)
            gerr = max(gerr, rel_err(np.where(f, x, 0.0), np.where(f, r, 0.0)))
            # per entry too; dL_dverts' floor is the reference formula's own float noise, see fuzz_case_noise.py
            if k != "verts" and not elementwise_close(np.where(f, x, 0.0), np.where(f, r, 0.0)):
                print("   per-entry mismatch:", k, flush=True); ok = False
        desc = f"tri seed {seed} B {B} H {H} W {W} P {P} F {F} R {R} rows {rows}"
    else:
        npts = int(rng.randint(20, 500))
        d = _delaunay(seed, npts, B, H, W)
        sc = oracle.scene_from_module_inputs(d, H, W)
        ocolor, odepth, oactive, ost = oracle.tet_forward(sc)
        args = c_args(d, dev, tet=True)
        out = _C.render_tets(*args, H, W, 0); th.cuda.synchronize()
        ex = lambda n, dt: _C.export(n, args, True, 0, out[3:7], H, W, dt).cpu().numpy()
        ok = np.array_equal(ex("first_face", th.int32), ost.get("first_face")) and np.array_equal(out[2].cpu().numpy(), oactive)
        # The march stops when expf(log_T) < T_EPS with log_T a sum of logf's: ocml's and glibc's expf/logf differ in the
        # last ulp (as CUDA's do from both), so a pixel whose transmittance lands within ~1e-7 (relative) of T_EPS may
        # march one face more or less.  Such pixels are counted, checked to be exactly that, and left out of the errors.
        lf_g, lf_o = ex("last_face", th.int32).reshape(-1), ost.get("last_face").reshape(-1)
        odd = np.nonzero(lf_g != lf_o)[0]
        if len(odd):
            near = np.full(len(odd), np.inf)
            for name in ("final_T", "final_prev_T"):
                for arr in (ex(name, th.float32).reshape(-1), ost.get(name).reshape(-1)):
                    near = np.minimum(near, np.abs(np.exp(arr[odd].astype(np.float64)) - 1e-4))
            ok = ok and bool((near <= 1e-9).all())
            n_marginal += len(odd)
        keep = np.ones(B * H * W, bool); keep[odd] = False
        kc = np.repeat(keep.reshape(B, 1, H, W), 3, axis=1)
        ferr = float(max(np.abs(np.where(kc, out[0].cpu().numpy() - ocolor, 0)).max(),
                         np.abs(np.where(keep.reshape(odepth.shape), out[1].cpu().numpy() - odepth, 0)).max()))
        og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
        g = _C.render_tets_backward(*args, gc.to(dev), gd.to(dev), *out[3:7]); th.cuda.synchronize()
        gerr = max(rel_err(got.cpu().numpy(), og[k]) for got, k in zip(g, ("verts_color", "faces_opacity")))
        # the same step again: its backward runs on the forward's march sequence (the first one of a view configuration re-marches)
        out2 = _C.render_tets(*args, H, W, 0)
        g2 = _C.render_tets_backward(*args, gc.to(dev), gd.to(dev), *out2[3:7]); th.cuda.synchronize()
        longest, cap = (int(x) for x in ex("tet_seq", th.int32).view(np.uint32)[:2])
        l2, cap2 = (int(x) for x in _C.export("tet_seq", args, True, 0, out2[3:7], H, W, th.int32).cpu().numpy().view(np.uint32)[:2])
        ok = ok and th.equal(out2[0], out[0]) and l2 == longest
        n_seq = globals().get("n_seq", 0) + (1 if 0 < l2 <= cap2 else 0); globals()["n_seq"] = n_seq
        gerr = max(gerr, max(rel_err(got.cpu().numpy(), og[k]) for got, k in zip(g2, ("verts_color", "faces_opacity"))))
        desc = f"tet seed {seed} B {B} H {H} W {W} npts {npts}"
    bad = (not ok) or not (ferr <= 1e-5) or not (gerr <= 1e-4)
    print(f"{'FAIL' if bad else 'ok  '} {desc} fwd_err {ferr:.2e} grad_err {gerr:.2e}", flush=True)
    if bad: sys.exit(1)
    seed += 1; ncase += 1
print(f"{globals().get('n_seq', 0)} tet cases ran their second backward on the march sequence", flush=True)
print(f"{ncase} cases, no mismatch ({n_marginal} tet pixels within 1e-9 of T_EPS marched one face more or less: libm ulp)", flush=True)
