#!/usr/bin/env python3
"""bench.py -- fwd+bwd Mpixels/s of the tri renderer at 1920x1080, 500k triangles (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config C1|C2|C3|C4|C5]

(--config C3 is the tet renderer's step -- render_tets + render_tets_backward on the Kuhn lattice of BASELINE configs[2] --
with its own roofline / cpu_baseline objects; the default and the headline metric are C4.)

A step = one forward + one backward of the hot path (`_C.render_tris` + `_C.render_tris_backward`,
i.e. the C ABI of libdmesh_renderer_hip.so) over one synthetic "layered sheets" scene (C4:
16 sheets x 126^2 vertices = 500 000 triangles, one 1920x1080 view), inputs resident in HBM.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL) -- started by a launcher (torchrun sets
WORLD_SIZE) or, when there is none, by this script itself (launch_ranks: N fresh child processes); the image is sharded by
work-balanced bands of tile rows, every rank renders and back-propagates its band, and the five
gradient tensors are summed with ONE all-reduce over a flattened fp32 buffer (strong scaling).

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, HIP-event timed inside
the timed region on the launch stream) and `cpu_baseline` (the CPU oracle = a port of the reference
algorithm, timed on this box's host cores; N == 1 only).  The oracle is used only as checker/baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# host threads of the cpu_baseline leg: the 1-GPU share of the box (16), set before any OpenMP runtime loads
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

import numpy as np
import torch as th
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C4", help="scene config (C1, C2, C4, C5: tri; C3: tet); the metric is quoted on C4")
    ap.add_argument("--opacity", default=None, help="lo,hi: face opacities U(lo, hi) instead of the config's (0.5,0.95 = the early-out scene)")
    ap.add_argument("--no-early-out", action="store_true", help="skip the second, early-termination record (tri, N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tet", action="store_true", help="skip the tet_c3 / tri_c2 / tri_c5 sub-records of the default (C4) run")
    ap.add_argument("--sync", action="store_true", help="time default (waiting) calls instead of asynchronous ones")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage timing table to stderr")
    ap.add_argument("--settle-ms", type=float, default=100.0, help="untimed steps before the warm-up, for about this long (0: none)")
    ap.add_argument("--dry-run", action="store_true", help="launch + rendezvous only (gloo, no GPU): prints n_gpus")
    ap.add_argument("--partition", default="auto", choices=("auto", "bands", "view_bands"),
                    help="N > 1: tile-row bands of all views, or (view, band) pairs (auto: the latter when B > 1 and N %% B == 0)")
    ap.add_argument("--emulate-rank", default=None, metavar="R/N",
                    help="one process, no collective: time rank R's tile-row band of an N-rank run (for rocprofv3 traces of the per-rank work)")
    return ap.parse_args()


def launch_ranks(n: int, argv, script=None, timeout_s: float = 3000.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, rendezvous on 127.0.0.1) and wait for them.  Called before anything in this
    process has touched the GPU (no torch.cuda.is_available(), no HIP call): the children are new processes, this
    one never re-executes itself and never initialises the device.  Returns the exit code for the parent: 0 only if
    every rank exited 0; the first failing rank takes the others down (they would hang in a collective)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc, t_end = 0, time.time() + timeout_s
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
        if (rc != 0 or time.time() > t_end) and live:
            for p in live:  # exactly the PIDs started above
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def dry_run(a, world: int, rank: int):
    """--dry-run: the N-rank launch and rendezvous without a GPU (gloo on the CPU): every rank joins the process
    group, the ranks are counted with one all-reduce and rank 0 prints {"n_gpus": N}.  tests/test_bench_launch_cpu.py."""
    n = 1
    if world > 1:
        dist.init_process_group(backend="gloo")
        t = th.ones(1, dtype=th.int64)
        dist.all_reduce(t)
        n = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": n, "world_size_env": world, "gpus_flag": a.gpus}), flush=True)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    backend = os.environ.get("DMR_DIST_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher around us: become one.  Nothing above has initialised the GPU (device_count() does not).
        if not a.dry_run and backend == "nccl" and th.cuda.device_count() < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} needs {a.gpus} HIP devices, found {th.cuda.device_count()} "
                             "(DMR_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if a.dry_run:
        return dry_run(a, world, rank)
    if not th.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # DMR_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the sharded path on a 1-GPU box;
    # the driver's multi-GPU runs use nccl (= RCCL), one rank per GPU
    if backend == "nccl" and th.cuda.device_count() < world and world > 1:
        raise SystemExit(f"{world} ranks need {world} HIP devices, found {th.cuda.device_count()}")
    local_rank = local_rank % max(1, th.cuda.device_count()) if backend != "nccl" else local_rank
    th.cuda.set_device(local_rank)
    dev = th.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    n_gpus = world

    from dmesh_renderer_amd import _C, scenes
    from dmesh_renderer_amd.scenes import c_args, max_abs_err, rel_err, upstream_grads
    from dmesh_renderer_amd.sharding import (balanced_bands, row_work_from_ranges, segment_cost, view_row_work_from_ranges,
                                             view_shares)

    cfg = scenes.CONFIGS[a.config]
    tet = cfg.kind == "tet"
    over = {}
    if a.opacity:
        lo, hi = (float(x) for x in a.opacity.split(","))
        over["opacity"] = (lo, hi)
    d = scenes.make(a.config, **over)
    B, H, W = cfg.B, cfg.H, cfg.W
    args = c_args(d, dev, tet=tet)
    gc_cpu, gd_cpu = upstream_grads(B, H, W)
    gc, gd = gc_cpu.to(dev), gd_cpu.to(dev)
    P, F = d["verts"].shape[0], d["faces"].shape[0]
    gy, gx = (H + 15) // 16, (W + 15) // 16
    GRADS = ("verts_color", "faces_opacity") if tet else ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")

    def forward(rows=(0, 0), fill=True, args=args, B=B):
        """-> (num_rendered, images..., four scratch buffers)"""
        if tet:
            o = _C.render_tets(*args, H, W, 0, rows=rows)
            return (None,) + tuple(o)
        return _C.render_tris(*args, H, W, rows=rows, fill_outside=fill)

    # one untimed full forward: scene statistics, and the work-balanced tile-row band of this rank
    out = forward()
    bufs = out[-4:]
    R_returned = None if tet else int(out[0])  # num_rendered as a default (waiting) call returns it
    ranges = _C.export("ranges", args, tet, 0 if tet else out[0], bufs, H, W, th.int32).cpu().numpy().reshape(-1, 2).astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    R_full = int(lens.sum())
    stats = {"tiles": int(lens.size), "tiles_busy": int((lens > 0).sum()),
             "list_len_mean": round(float(lens[lens > 0].mean()) if (lens > 0).any() else 0.0, 1), "list_len_max": int(lens.max())}
    if tet:  # S = marched (pixel, face) pairs
        stats["marched_pairs"] = int(_C.export("n_contrib", args, True, 0, bufs, H, W, th.int32).sum().item())
    else:    # blended (pixel, face) pairs: what the backward's record stream holds
        tile_hits = _C.export("tile_hits", args, False, out[0], bufs, H, W, th.int32).cpu().numpy().astype(np.int64)
        stats["blended_pairs"] = int(tile_hits.sum())
    rows = (0, 0)
    emu = None
    if a.emulate_rank:
        emu = tuple(int(x) for x in a.emulate_rank.split("/"))
        assert world == 1 and 0 <= emu[0] < emu[1]
    nparts, part = (emu[1], emu[0]) if emu else (world, rank)
    # (view, band) segments with several views (sharding.view_shares): the rows of all views, view after view, cut into N shares
    # of equal cost; a rank renders its segments with B = 1 tensors -- it projects, bins and scatters ONE view's faces per call
    segs = None
    if (world > 1 or emu) and not tet and B > 1 and a.partition != "bands":
        segs = view_shares(view_row_work_from_ranges(ranges, B, gy, gx, tile_hits=tile_hits), nparts, segment_cost(F, True))[part]
        segs = segs or [(0, gy, gy)]  # a rank without rows still steps (and joins the collective): an empty band
    elif a.partition == "view_bands" and (world > 1 or emu):
        raise SystemExit("--partition view_bands needs a tri config with B > 1")
    seg_args = []
    if segs is not None:
        for v, r0, r1 in segs:
            sa = list(args)
            for i in range(5, 11):  # mv, proj, their inverses, verts_depth, faces_intense: this view's
                sa[i] = args[i][v:v + 1].contiguous()
            seg_args.append((v, (r0, r1), sa, gc[v:v + 1].contiguous(), gd[v:v + 1].contiguous()))
    elif world > 1 or emu:
        rows = balanced_bands(row_work_from_ranges(ranges, B, gy, gx, tile_hits=None if tet else tile_hits), nparts)[part]
        if rows[1] <= rows[0]:
            rows = (gy, gy)  # an empty band ((0, 0) would mean "all rows")
    del out, bufs

    # the all-reduce payload: [3P | 3P | F | B*P | B*F] (tet: [3P | F]) of ALL views
    flat = th.empty(3 * P + F if tet else 6 * P + F + B * (P + F), dtype=th.float32, device=dev)
    flat1 = th.empty(7 * P + 2 * F, dtype=th.float32, device=dev) if segs is not None else None

    def step_segments():
        """This rank's (view, band) segments, one B = 1 forward + backward each; their gradients into the all-views payload:
        the shared part summed over the segments, each segment's per-view rows in its view's place, the other views' rows zero
        (the all-reduce overwrites the payload with the sums: cleared every step)."""
        sh = 6 * P + F
        flat[sh:].zero_()
        o = g = None
        for i, (v, rws, sa, gcv, gdv) in enumerate(seg_args):
            o = forward(rws, fill=False, args=sa, B=1)
            g = _C.render_tris_backward(*sa, gcv, gdv, o[0], *o[-4:], rows=rws, flat_out=flat1)
            if i == 0:
                flat[:sh].copy_(flat1[:sh])
            else:
                flat[:sh].add_(flat1[:sh])
            flat[sh + v * P:sh + (v + 1) * P].add_(flat1[sh:sh + P])
            flat[sh + B * P + v * F:sh + B * P + (v + 1) * F].add_(flat1[sh + P:])
        if world > 1:
            dist.all_reduce(flat)
        return o, g

    def step():
        if segs is not None:
            return step_segments()
        o = forward(rows, fill=False)  # a rank only owns the rows of its band
        kw = {}
        if world > 1 or emu:  # the gradients land back to back in `flat` (views are returned): ONE collective, no concatenation
            kw = dict(rows=rows, flat_out=flat)
        if tet:
            g = _C.render_tets_backward(*args, gc, gd, *o[-4:], **kw)
        else:
            g = _C.render_tris_backward(*args, gc, gd, o[0], *o[-4:], **kw)
        if world > 1:
            dist.all_reduce(flat)
        return o, g

    def collect():
        ms, cnt = _C.profile_collect()
        return np.array(ms), np.array(cnt)

    def barrier():
        if world > 1:
            dist.barrier()
        th.cuda.synchronize()

    # The dominant kernel is HIP-event timed INSIDE the timed region (2 events per launch on the launch stream).
    # Which of the three big kernels that is comes from the warm-up steps: every pair of events costs a few
    # microseconds of the step being measured, so only one stage is timed there.
    cand = (_C.STAGE_TET_FIRST, _C.STAGE_TET_FORWARD, _C.STAGE_TET_BACKWARD) if tet else \
        (_C.STAGE_TRI_FORWARD, _C.STAGE_TRI_BACKWARD, _C.STAGE_TRI_BACKWARD_HITS)
    mask = 0
    for c in cand:
        mask |= 1 << c
    # Settling phase, before the W warm-up steps and the K timed ones: the same step, untimed, for ~0.1 s (a fresh process on
    # an idle chip measures 2-3 % slower during its first ~50 ms: clocks, TLBs; `--warmup 5` alone is 1.6 ms of C4).  Nothing in
    # the timed region changes; the number of settling steps is in the JSON line (`settle_steps`).  Every rank runs the same count.
    step(); th.cuda.synchronize()
    t0 = time.perf_counter(); step(); th.cuda.synchronize()
    settle = int(min(2000, max(0, a.settle_ms * 1e-3 / max(time.perf_counter() - t0, 1e-5))))
    if world > 1:
        ss = th.tensor([settle], dtype=th.int64, device=dev)
        dist.broadcast(ss, 0)
        settle = int(ss.item())
    for _ in range(settle):
        step()
    th.cuda.synchronize()
    nwarm = max(1, a.warmup)
    for i in range(nwarm):
        if i == min(1, nwarm - 1):
            _C.profile_enable(mask)  # not the very first step: it sizes its buffers without an estimate and may run stages twice
        step()
    th.cuda.synchronize()
    _C.profile_enable(0)
    wms, wcnt = collect()
    dom = max(cand, key=lambda i: wms[i] / max(1, wcnt[i]))
    if world > 1:  # every rank times the same stage
        dd = th.tensor([dom], dtype=th.int64, device=dev)
        dist.broadcast(dd, 0)
        dom = int(dd.item())
    # The timed steps are ASYNCHRONOUS calls (C ABI DMR_FLAG_ASYNC): buffers sized from the warm-up steps'
    # estimates, no device->host size read-back inside a step (the reference waits for 4 bytes in every forward,
    # rasterizer_impl.cu:287-292).  The device raises a sticky flag if a step outgrew its buffers; it is read after the timed
    # region, and the measurement is then repeated with default (waiting) calls.  --sync times default calls from the start.
    # The dominant kernel's HIP events are recorded in every PROF_EVERY-th timed step only: an event record between two kernels
    # costs the device 2-7 us (profiles/r03/sync_poll_vs_event_c4.txt: one record per step, 0.2947 -> 0.2993 ms), two per step
    # would be a few per cent of the step being measured.  Still live, on the launch stream, inside the timed region.
    PROF_EVERY = 4

    def timed(asynchronous):
        _C.overflowed()  # clear
        _C.set_async(asynchronous)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            _C.profile_enable((1 << dom) if i % PROF_EVERY == 0 else 0)
            o, g = step()
        barrier()
        dt = time.perf_counter() - t0
        _C.profile_enable(0)
        _C.set_async(False)
        return o, g, dt, bool(_C.overflowed())

    host_sync = "per call (default)" if a.sync else "none in the timed steps (asynchronous calls, overflow flag read afterwards)"
    o, g, dt, overflowed = timed(not a.sync)
    if overflowed:
        collect()
        host_sync = "per call (default): the asynchronous steps overflowed their buffers and were discarded"
        o, g, dt, _ = timed(False)
    if world > 1:
        tt = th.tensor([dt], dtype=th.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    value = B * W * H * a.steps / dt / 1e6
    # `sync`: the same K steps with DEFAULT (waiting) calls, taken right behind the timed region, in the same allocator state
    # (behind the early-out scene's different buffer sizes the same steps measure 1-5 % slower: where the caching allocator
    # places the work buffer) -- what a user of the drop-in TriRenderer / `_C` gets without opting into asynchronous calls.
    ms_main, cnt_main = collect()
    # N > 1: the all-reduced gradients of the last timed step against ONE rank rendering the whole frame alone (untimed; every
    # rank does it -- the collective above needs no partner here): what the sharding and the collective must not change
    sharded_err = None
    if world > 1:
        summed = flat.clone()
        fo = forward()
        full = th.empty_like(flat)
        if tet:
            _C.render_tets_backward(*args, gc, gd, *fo[-4:], flat_out=full)
        else:
            _C.render_tris_backward(*args, gc, gd, fo[0], *fo[-4:], flat_out=full)
        sizes = (3 * P, F) if tet else (3 * P, 3 * P, F, B * P, B * F)
        sharded_err, at = 0.0, 0
        for n in sizes:
            sharded_err = max(sharded_err, rel_err(summed[at:at + n].cpu().numpy(), full[at:at + n].cpu().numpy()))
            at += n
        del summed, full, fo
    sync_rec = None
    if world == 1 and not emu and rank == 0 and not a.sync and not a.opacity and not overflowed:
        for _ in range(max(3, a.warmup)):  # a few default steps first: the timed ones before were asynchronous
            step()
        th.cuda.synchronize()
        _o, _g, dts, _ = timed(False)
        collect()  # (its own event samples are not the roofline's)
        sync_rec = {"ms_per_step": round(dts / a.steps * 1e3, 4), "value": round(B * W * H * a.steps / dts / 1e6, 2),
                    "host_sync": "per call (default)"}
        del _o, _g

    # roofline of the dominant kernel.  Algorithmic bytes per launch (rays fused):
    #   k_tri_forward / k_tri_backward_pix : 132 B per list entry (4 B id + 128 B face record) + 28 B per pixel
    #   k_tri_backward_hits                : 184 B per list entry (23 fp32 read-modify-writes per (tile, face))
    #   k_tet_first_intersect              : 52 B per list entry + 8 B per pixel (first face / tet)
    #   k_tet_forward / k_tet_backward     : the packed mesh records once + 4 B per marched pair S (the march sequence) +
    #       the per-pixel state and images (see below; SURVEY 8(d)'s 508 / 548 B per step are the reference's L2-served gathers)
    # (the hit-record stream between the two tri backward kernels is this design's own traffic, not algorithmic)
    if segs is not None:  # this rank's segments
        R, npix_band = 0, 0
        for v, rws, sa, _, _ in seg_args[-1:]:  # (the scratch buffers of the last segment's forward are the ones still alive)
            br = _C.export("ranges", sa, False, o[0], o[-4:], H, W, th.int32).cpu().numpy().reshape(-1, 2).astype(np.int64)
            R += int((br[:, 1] - br[:, 0]).sum())
        for v, (r0, r1), _, _, _ in seg_args:
            npix_band += W * max(0, min(H, r1 * 16) - r0 * 16)
        if len(seg_args) > 1:  # the roofline's bytes are per launch: the last segment's
            v, (r0, r1), _, _, _ = seg_args[-1]
            npix_band = W * max(0, min(H, r1 * 16) - r0 * 16)
    elif world > 1 or emu:  # this rank's band
        br = _C.export("ranges", args, tet, 0 if tet else o[0], o[-4:], H, W, th.int32).cpu().numpy().reshape(-1, 2).astype(np.int64)
        R = int((br[:, 1] - br[:, 0]).sum())
        npix_band = B * W * max(0, min(H, rows[1] * 16) - rows[0] * 16)
    else:
        R, npix_band = R_full, B * W * H
    if tet:
        # The march kernels' algorithmic bytes are this design's compulsory HBM traffic: the packed
        # march records once per launch (128 B per face + 224 B per tet: the mesh stays in L2, its per-step gathers are
        # not HBM traffic), the march sequence (4 B per marched pair S, written by the forward, read by the backward)
        # and the per-pixel state / images.  SURVEY 8(d)'s 508 / 548 B per step price the REFERENCE's gathers; against
        # the HBM peak they gave a "fraction" of 2.3 (round 2), which is not a roofline.
        S = int(_C.export("n_contrib", args, True, 0, o[-4:], H, W, th.int32).sum().item())
        T = int(d["tets"].shape[0])
        mesh = 128.0 * F + 224.0 * T
        alg = {_C.STAGE_TET_FIRST: 52.0 * R + 8.0 * npix_band,
               _C.STAGE_TET_FORWARD: mesh + 4.0 * S + (8 + 21 + 20) * npix_band,   # first face/tet in; state 21 B, colour/depth/active 20 B out
               _C.STAGE_TET_BACKWARD: mesh + 4.0 * S + (21 + 8 + 16) * npix_band + 2 * 4.0 * (3 * P + F)}  # state, first face/tet, dL_dpix in; gradients RMW
    else:
        alg = {_C.STAGE_TRI_FORWARD: 132.0 * R + 28.0 * npix_band, _C.STAGE_TRI_BACKWARD: 132.0 * R + 28.0 * npix_band,
               _C.STAGE_TRI_BACKWARD_HITS: 184.0 * R}
    ms, cnt = ms_main, cnt_main
    dom_name = _C.stage_name(dom)
    dom_ms = ms[dom] / max(1, cnt[dom])
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0

    def committed(kind):
        """A per-kernel figure measured offline with rocprofv3 on this workload and committed under profiles/ (newest round first)."""
        for rnd in ("r03", "r02", "r01"):
            path = os.path.join(ROOT, "profiles", rnd, f"{kind}_{a.config.lower()}.json")
            if os.path.exists(path):
                j = json.load(open(path))
                rec = j.get("dmr::" + dom_name)
                # the stage k_tet_backward is two launches of which the device runs one (dmr_kernels.hpp): their counters add up
                other = j.get("dmr::k_tet_backward_seq") if dom_name == "k_tet_backward" else None
                if rec and other:
                    rec = {k: (other.get(k, 0.0) if k == "valu_lane_util" else rec.get(k, 0.0) + other.get(k, 0.0))
                           for k in set(rec) | set(other) if isinstance(rec.get(k, other.get(k)), (int, float))}
                return rec or other, f"profiles/{rnd}/{kind}_{a.config.lower()}.json"
        return None, None

    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes, scripts/prof_traffic.sh)
    traffic, valu = None, None
    if world == 1 and not a.opacity and not emu:
        tj, _ = committed("traffic")
        traffic = (tj or {}).get("hbm_bytes_per_launch")
        # secondary figure: VALU issue.  SQ_INSTS_VALU wave-instructions per launch (committed PMC pass) x 2.25
        # SIMD cycles each (profiles/r02/valu_issue.txt: what a SIMD sustains with >= 2 waves ready; DPP / integer-multiply /
        # packed forms cost twice that, so this is a lower bound of the issue time) over 1024 SIMDs at 2.4 GHz, against the
        # kernel's measured duration
        pj, src = committed("pmc")
        if pj and pj.get("SQ_INSTS_VALU") and dom_ms > 0:
            insts, cpi = float(pj["SQ_INSTS_VALU"]), 2.25
            valu = {"insts": insts, "cycles_per_inst": cpi, "issue_ms": round(insts * cpi / 1024 / 2.4e9 * 1e3, 4),
                    "frac": round(insts * cpi / 1024 / 2.4e9 / (dom_ms * 1e-3), 4),
                    # active lanes per issued VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), one counter pass
                    "lane_util": round(float(pj["valu_lane_util"]), 4) if pj.get("valu_lane_util") else None, "source": src}
            # ... and how busy the VALUs were by the counters' own account: rocprof's VALUBusy with the kernel's busy CU-cycles as the
            # time base, SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES (one counter pass, scripts/prof_valu_mix.sh).  "frac" above prices every
            # instruction at the 2.25 cycles of a plain fp32 add / mul; most of these kernels' instructions are of the 4.5-cycle
            # kind (compares, selects, integer multiply-adds, fma with a register-bank conflict: profiles/r03/valu_classes.txt).
            mj, msrc = committed("valu_mix")
            if mj and mj.get("SQ_ACTIVE_INST_VALU") and mj.get("SQ_BUSY_CU_CYCLES"):
                valu["busy"] = round(float(mj["SQ_ACTIVE_INST_VALU"]) / float(mj["SQ_BUSY_CU_CYCLES"]), 4)
                valu["busy_source"] = msrc
    note = ("HBM is not the bound: the compositing kernels are VALU-bound -- SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES = 0.93 (forward), "
            "0.76 (per-pixel backward), 0.74 (hit-parallel backward) at C4 -- with most instructions of the 4.5-cycle kind (integer, "
            "compare, select), and a tile's serial chain of barrier-separated phases on top (DESIGN.md section 4)")
    if tet:
        note = ("algorithmic bytes = this design's compulsory HBM traffic (packed mesh records once, 4 B per marched pair of the march "
                "sequence, per-pixel state and images); the per-step record gathers are served by L2 (the mesh is 6.5 MB) and are not "
                "HBM traffic.  The march is bound by its dependent-gather chain per step (forward) and by VALU issue (backward), "
                "not by HBM (DESIGN.md section 5b)")
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_ms": round(dom_ms, 4), "launches_timed": int(cnt[dom]), "algorithmic_bytes": alg[dom], "valu": valu, "note": note}

    # full per-stage table (separate, untimed-for-value pass)
    _C.profile_enable(0xFFFFFFFF)
    for _ in range(min(10, a.steps)):
        step()
    th.cuda.synchronize()
    _C.profile_enable(0)
    sms, scnt = collect()
    stages = {_C.stage_name(i): round(float(sms[i] / scnt[i]), 4) for i in range(_C.NUM_STAGES) if scnt[i]}
    if a.stages and rank == 0:
        print("per-stage avg ms:", json.dumps(stages), file=sys.stderr)

    # second record: the same scene with opacities U(0.5, 0.95) -- pixels terminate after a few faces, which
    # exercises the early-out of the compositing loops
    early = None
    if world == 1 and not tet and not a.opacity and not a.no_early_out and not emu and rank == 0:
        d2 = scenes.make(a.config, opacity=(0.5, 0.95))
        args2 = c_args(d2, dev)
        def step2():
            o2 = _C.render_tris(*args2, H, W)
            return o2, _C.render_tris_backward(*args2, gc, gd, o2[0], *o2[3:7])
        for _ in range(max(1, a.warmup)):
            step2()
        th.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            o2, _g2 = step2()
        th.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        early = {"opacity": [0.5, 0.95], "ms_per_step": round(dt2 / a.steps * 1e3, 4), "value": round(B * W * H * a.steps / dt2 / 1e6, 2),
                 "num_rendered": int(o2[0]),
                 "blended_pairs": int(_C.export("tile_hits", args2, False, o2[0], o2[3:7], H, W, th.int32).long().sum().item()),
                 "pixels_terminated": int((_C.export("final_T", args2, False, o2[0], o2[3:7], H, W, th.float32) < 1e-4).sum().item())}
        del d2, args2, o2, _g2

    # Sub-records the driver-run line carries next to the headline (untimed for `value`):
    #   sync   -- the same K steps with DEFAULT (waiting) calls: what a user of the drop-in TriRenderer / `_C` gets without
    #             opting into asynchronous calls (the headline is the opt-in mode)
    #   tet_c3 -- the tet renderer's step on BASELINE configs[2] (render_tets + render_tets_backward, default calls) with its
    #             parity against the oracle
    tet_rec = None
    if world == 1 and not emu and rank == 0 and a.config == "C4" and not a.opacity and not a.no_tet:
        tcfg = scenes.CONFIGS["C3"]
        td = scenes.make("C3")
        targs = c_args(td, dev, tet=True)
        tgc_cpu, tgd_cpu = upstream_grads(tcfg.B, tcfg.H, tcfg.W)
        tgc, tgd = tgc_cpu.to(dev), tgd_cpu.to(dev)
        def tstep():
            to = _C.render_tets(*targs, tcfg.H, tcfg.W, 0)
            return to, _C.render_tets_backward(*targs, tgc, tgd, *to[3:7])
        for _ in range(max(3, a.warmup)):
            tstep()
        th.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            to, tg = tstep()
        th.cuda.synchronize()
        dtt = time.perf_counter() - t1
        tet_rec = {"workload": f"C3: {tcfg.name}, Kuhn lattice seed 0, B={tcfg.B}", "ms_per_step": round(dtt / a.steps * 1e3, 4),
                   "value": round(tcfg.B * tcfg.W * tcfg.H * a.steps / dtt / 1e6, 2), "unit": "Mpixels/s", "host_sync": "per call (default)",
                   "marched_pairs": int(_C.export("n_contrib", targs, True, 0, to[3:7], tcfg.H, tcfg.W, th.int32).sum().item()),
                   "march_sequence": dict(zip(("longest_march", "capacity_steps"),
                                              (int(x) for x in _C.export("tet_seq", targs, True, 0, to[3:7], tcfg.H, tcfg.W, th.int32).cpu().numpy().view(np.uint32)[:2])))}
        if not a.no_cpu_baseline:
            from oracle import oracle as O  # checker only
            O.build()
            tsc = O.scene_from_module_inputs(td, tcfg.H, tcfg.W)
            tocolor, todepth, toactive, tost = O.tet_forward(tsc)
            tog = O.tet_backward(tsc, tost, tgc_cpu.numpy(), tgd_cpu.numpy())
            tet_rec["fwd_max_abs_err"] = float(max(np.abs(to[0].cpu().numpy() - tocolor).max(), np.abs(to[1].cpu().numpy() - todepth).max()))
            tet_rec["grad_max_norm_err"] = float(max(rel_err(t.cpu().numpy(), tog[k]) for t, k in zip(tg, ("verts_color", "faces_opacity"))))
            tet_rec["active_equal"] = bool(np.array_equal(to[2].cpu().numpy(), toactive))
        del td, targs, to, tg

    # ... and the other two tri configurations of BASELINE.json, so that the driver-run line has them (VERDICT r02 "missing" 4):
    # C2 (100k triangles, 800 x 800) with its parity, C5 (2M triangles, 4096 x 4096, 4 views -- the 8-GPU configuration, here on
    # this one GPU) without the oracle (minutes on the host; tests/test_fullsize_gpu.py::test_c5_matches_oracle holds its parity)
    others = {}
    if world == 1 and not emu and rank == 0 and a.config == "C4" and not a.opacity and not a.no_tet:
        for name, nsteps in (("C2", a.steps), ("C5", max(3, min(a.steps, 10)))):
            ocfg = scenes.CONFIGS[name]
            od = scenes.make(name)
            oargs = c_args(od, dev)
            ogc_cpu, ogd_cpu = upstream_grads(ocfg.B, ocfg.H, ocfg.W)
            ogc, ogd = ogc_cpu.to(dev), ogd_cpu.to(dev)
            def ostep():
                oo = _C.render_tris(*oargs, ocfg.H, ocfg.W)
                return oo, _C.render_tris_backward(*oargs, ogc, ogd, oo[0], *oo[3:7])
            for _ in range(3):
                ostep()
            th.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nsteps):
                oo, og_ = ostep()
            th.cuda.synchronize()
            dto = time.perf_counter() - t1
            rec = {"workload": f"{name}: {ocfg.name}, layered sheets seed 0, B={ocfg.B}", "steps": nsteps, "ms_per_step": round(dto / nsteps * 1e3, 4),
                   "value": round(ocfg.B * ocfg.W * ocfg.H * nsteps / dto / 1e6, 2), "unit": "Mpixels/s", "host_sync": "per call (default)",
                   "num_rendered": int(oo[0])}
            if name == "C2" and not a.no_cpu_baseline:
                from oracle import oracle as O  # checker only
                O.build()
                osc = O.scene_from_module_inputs(od, ocfg.H, ocfg.W)
                ooc, ood, oost = O.tri_forward(osc)
                oog = O.tri_backward(osc, oost, ogc_cpu.numpy(), ogd_cpu.numpy())
                rec["fwd_max_abs_err"] = float(max(np.abs(oo[1].cpu().numpy() - ooc).max(), np.abs(oo[2].cpu().numpy() - ood).max()))
                rec["grad_max_norm_err"] = float(max(rel_err(t.cpu().numpy(), oog[k]) for t, k in zip(og_, GRADS)))
                rec["num_rendered_equal"] = bool(int(oo[0]) == oost.num_rendered)
            others[name.lower()] = rec
            del od, oargs, oo, og_, ogc, ogd
            th.cuda.empty_cache()

    cpu_baseline = None
    parity = {}
    if world == 1 and not a.no_cpu_baseline and not emu and rank == 0:
        from oracle import oracle as O  # checker / reported baseline only
        O.build()
        sc = O.scene_from_module_inputs(d, H, W)
        # a bounded sample: whole passes over the same workload until ~10 s of CPU work (3..40 passes; one pass for the
        # minutes-long C5); the first pass also pays page faults and thread start-up, hence the median
        min_reps, max_reps, budget_s = (3, 40, 10.0) if a.config in ("C1", "C2", "C3", "C4") else (1, 1, 0.0)
        times = []
        while len(times) < min_reps or (len(times) < max_reps and sum(times) < budget_s):
            t1 = time.perf_counter()
            if tet:
                ocolor, odepth, oactive, ost = O.tet_forward(sc)
                og = O.tet_backward(sc, ost, gc_cpu.numpy(), gd_cpu.numpy())
            else:
                ocolor, odepth, ost = O.tri_forward(sc)
                og = O.tri_backward(sc, ost, gc_cpu.numpy(), gd_cpu.numpy())
            times.append(time.perf_counter() - t1)
        cdt = sorted(times)[len(times) // 2]
        reps = len(times)
        cores = int(O.lib().dmro_num_threads())
        cpu_baseline = {"value": round(B * W * H / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                        "sample": f"median of {reps} fwd+bwd passes over the full {a.config} workload ({cdt:.2f} s each, "
                                  f"{sum(times):.1f} s in all, OpenMP {cores} threads)"}
        color, depth = (o[1], o[2])
        parity["fwd_max_abs_err"] = float(max(np.abs(color.cpu().numpy() - ocolor).max(), np.abs(depth.cpu().numpy() - odepth).max()))
        # grad_max_abs_err: plain max |g - g_oracle| over all gradient tensors; grad_max_norm_err: each tensor's
        # max-abs error over max(1, max-abs of the oracle's tensor), SURVEY 8(d)'s definition of the metric's second half
        parity["grad_max_abs_err"] = float(max(max_abs_err(t.cpu().numpy(), og[k]) for t, k in zip(g, GRADS)))
        parity["grad_max_norm_err"] = float(max(rel_err(t.cpu().numpy(), og[k]) for t, k in zip(g, GRADS)))
        if tet:
            parity["active_equal"] = bool(np.array_equal(o[3].cpu().numpy(), oactive))
        else:
            parity["num_rendered_equal"] = bool(R_returned == ost.num_rendered)

    if rank == 0:
        line = {
            "metric": "fwd+bwd Mpixels/sec @1080p, 500k tris; grad max-abs-err vs ref",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "settle_steps": settle,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": f"{a.config}: {cfg.name}, {'Kuhn lattice' if tet else 'layered sheets'} seed 0, B={B}"
                                        + (f", opacity U({a.opacity})" if a.opacity else ""),
                            "renderer": cfg.kind, "triangles": F, "verts": P, "image": [H, W], "num_rendered": int(R_full),
                            "host_sync": host_sync,
                            # what the ONE gradient all-reduce of an N-rank run carries: [3P | 3P | F | B*P | B*F] fp32 (tet: [3P | F])
                            "allreduce_payload_bytes": int(flat.numel() * 4),
                            "parallelism": (f"EMULATED rank {emu[0]} of {emu[1]}: "
                                            + (f"its (view, row_begin, row_end) segments {segs}" if segs is not None else f"its tile-row band {rows}")
                                            + ", no collective" if emu else
                                            "single GPU" if world == 1 else
                                            f"(view, tile-row band) segments of {B} views x{world} + 1 RCCL all-reduce" if segs is not None
                                            else f"tile-row bands x{world} + 1 RCCL all-reduce")}, **stats),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "stages_ms": stages, "early_out": early,
            "sharded_grad_max_norm_err": sharded_err,  # N > 1: all-reduced gradients vs one rank's whole-frame gradients
            "sync": sync_rec, "tet_c3": tet_rec, "tri_c2": others.get("c2"), "tri_c5": others.get("c5"),
        }
        if tet:
            line["config"]["tets"] = int(d["tets"].shape[0])
        line.update(parity)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
