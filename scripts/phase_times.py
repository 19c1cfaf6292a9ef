"""Where does a tile's serial chain spend its time?  In-kernel phase stamps of the three compositing kernels.

Runs one forward + backward of a BASELINE config on the ABLATION build (`python -m dmesh_renderer_amd.build --ablation`)
with DMR_ABLATE bit 4096: lane 0 of every wave of the first 3072 workgroups (tile_order: the longest lists) writes s_memtime
at every phase boundary of its first 6 chunks (dmr_tri.hip, DMR_STAMP).  The product library has no such code.

    python scripts/phase_times.py [--config C4] [--band R0 R1] [--json out.json]

--band R0 R1 renders only tile rows [R0, R1): with <= 256 busy tiles every workgroup has a CU of its own, which gives the
chain on an idle chip next to the chain under full load.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "dmesh_renderer_amd", "libdmesh_renderer_hip_ablation.so")
os.environ["DMR_LIBRARY"] = ABL
os.environ["DMR_ABLATE"] = str(4096 | int(os.environ.get("DMR_ABLATE", "0")))

import numpy as np  # noqa: E402
import torch as th  # noqa: E402

PH_KERNELS, PH_BLOCKS, PH_WAVES, PH_CHUNKS, PH_STAMPS = 3, 3072, 4, 6, 10
KERNELS = ("k_tri_forward", "k_tri_backward_pix", "k_tri_backward_hits")
PHASES = {
    0: [("barrier (top; waits for the slowest wave of the previous chunk)", 0, 1), ("stage records", 1, 2), ("barrier", 2, 3),
        ("rasterise (face-parallel)", 3, 4), ("barrier", 4, 5), ("walk + shade own bits", 5, 6)],
    1: [("barrier (top)", 0, 1), ("stage shading records, cut the forward's masks, count per face", 1, 2), ("barrier", 2, 3),
        ("scan counters (wave 0)", 5, 6), ("barrier", 6, 7), ("pad records", 7, 8), ("walk own bits, write records", 8, 9)],
    2: [("records + face gathers arrive", 0, 1), ("23 components of <= 4 pairs", 1, 2), ("segmented DPP scan", 2, 3),
        ("stage tails, table adds, face-row atomics", 3, 4)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4")
    ap.add_argument("--band", type=int, nargs=2, default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()

    from dmesh_renderer_amd import _C, build, scenes
    build.build(ablation=True)
    assert _C.library_path() == ABL, _C.library_path()
    lib = ctypes.CDLL(ABL)  # the same handle the binding holds
    lib.dmr_debug_phase.restype = ctypes.c_longlong
    lib.dmr_debug_phase.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]

    cfg = scenes.CONFIGS[a.config]
    dev = th.device("cuda:0")
    d = scenes.make(a.config)
    args = scenes.c_args(d, dev)
    B, H, W = cfg.B, cfg.H, cfg.W
    gen = th.Generator().manual_seed(1)
    gc = th.rand(B, 3, H, W, generator=gen).to(dev)
    gd = th.rand(B, 1, H, W, generator=gen).to(dev)
    kw = {"rows": tuple(a.band)} if a.band else {}

    def step():
        o = _C.render_tris(*args, H, W, **kw)
        g = _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7], **kw)
        return o, g

    for _ in range(3):
        step()
    th.cuda.synchronize()
    lib.dmr_debug_phase(0, None, 0, 1)
    lib.dmr_debug_phase(1, None, 0, 1)
    step()
    th.cuda.synchronize()
    ph = np.zeros((PH_KERNELS, PH_BLOCKS, PH_WAVES, PH_CHUNKS, PH_STAMPS), dtype=np.uint64)
    rt = np.zeros((PH_KERNELS, PH_BLOCKS, 2), dtype=np.uint64)
    assert lib.dmr_debug_phase(0, ph.ctypes.data, ph.nbytes, 0) == ph.nbytes
    assert lib.dmr_debug_phase(1, rt.ctypes.data, rt.nbytes, 0) == rt.nbytes
    ph = ph.astype(np.int64)
    rt = rt.astype(np.int64)

    out = {"config": a.config, "band": a.band, "kernels": {}}
    for kern in range(PH_KERNELS):
        busy = np.nonzero(rt[kern, :, 1] > 0)[0]
        if len(busy) == 0:
            continue
        # the clock the chip held: s_memtime ticks per 10 ns s_memrealtime tick, over whole workgroups
        if kern == 2:
            t0, t1 = ph[kern, busy, 0, 0, 8], ph[kern, busy, 0, 0, 7]
        else:
            t0 = ph[kern, busy, 0, 0, 0]
            last = np.array([ph[kern, b, 0, :, 9 if kern == 1 else 6].max() for b in busy])
            t1 = last
        wall = (rt[kern, busy, 1] - rt[kern, busy, 0]) * 10.0  # ns
        ok = (t1 > t0) & (wall > 0)
        mhz = float(np.median((t1[ok] - t0[ok]) / wall[ok]) * 1e3) if ok.any() else float("nan")
        rec = {"workgroups_stamped": int(len(busy)), "workgroup_us_mean": float(wall.mean() / 1e3),
               "workgroup_us_median": float(np.median(wall) / 1e3), "memtime_mhz_median": mhz, "phases": []}
        print(f"== {KERNELS[kern]}: {len(busy)} workgroups stamped, workgroup lifetime mean {wall.mean() / 1e3:.2f} us "
              f"(median {np.median(wall) / 1e3:.2f}), s_memtime at {mhz:.0f} MHz")
        nchunks = 0
        chunk_cycles = []
        for name, s0, s1 in PHASES[kern]:
            a0, a1 = ph[kern, busy][:, :, :, s0], ph[kern, busy][:, :, :, s1]  # [block, wave, chunk]
            valid = (a0 > 0) & (a1 >= a0)
            dur = np.where(valid, a1 - a0, 0)
            n = int(valid.sum())
            mean_all = float(dur.sum() / max(n, 1))
            # the slowest wave of each (block, chunk): what the following barrier waits for
            vmax = np.where(valid.any(axis=1), dur.max(axis=1), 0)
            nmax = int(valid.any(axis=1).sum())
            mean_max = float(vmax.sum() / max(nmax, 1))
            per_wave = [float(dur[:, w].sum() / max(int(valid[:, w].sum()), 1)) for w in range(PH_WAVES)]
            nchunks = max(nchunks, nmax)
            chunk_cycles.append(mean_max)
            rec["phases"].append({"phase": name, "cycles_mean": mean_all, "cycles_slowest_wave_mean": mean_max,
                                  "cycles_per_wave": per_wave, "samples": n})
            print(f"   {name:<66s} mean {mean_all:8.0f} cyc   slowest wave {mean_max:8.0f} cyc   per wave "
                  + " ".join(f"{x:7.0f}" for x in per_wave))
        # whole chunk: stamp 0 of chunk c+1 minus stamp 0 of chunk c (wave 0)
        s_first = 0
        c0 = ph[kern, busy][:, 0, :-1, s_first]
        c1 = ph[kern, busy][:, 0, 1:, s_first]
        v = (c0 > 0) & (c1 > c0)
        if v.any():
            cyc = float((c1 - c0)[v].mean())
            rec["chunk_cycles_mean"] = cyc
            rec["chunk_us_mean"] = cyc / mhz if mhz == mhz else None
            print(f"   one chunk / round, start to start (wave 0): {cyc:.0f} cycles = {cyc / mhz:.2f} us   ({int(v.sum())} samples)")
        if kern == 2:
            b0 = ph[kern, busy][:, :, 0, :]
            init = (b0[:, :, 9] - b0[:, :, 8]); endbar = (b0[:, :, 6] - b0[:, :, 5]); flush = (b0[:, :, 7] - b0[:, :, 6])
            rec["init_cycles"] = float(init.mean()); rec["end_barrier_cycles"] = float(endbar.mean()); rec["flush_cycles"] = float(flush.mean())
            print(f"   table init + pixel staging {init.mean():.0f} cyc, end barrier {endbar.mean():.0f} cyc, table flush {flush.mean():.0f} cyc")
        out["kernels"][KERNELS[kern]] = rec
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
