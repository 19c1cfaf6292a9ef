"""Builds dmesh_renderer_amd/libdmesh_renderer_hip.so (gfx950 only) with hipcc, in-tree.

    python -m dmesh_renderer_amd.build [--force] [--ablation]

`--ablation` builds a SEPARATE library, libdmesh_renderer_hip_ablation.so (-DDMR_ABLATION), whose kernels honour the
DMR_ABLATE environment variable (timing ablations of scripts/prof_ablate.sh, the forced-fallback switch of
tests/test_fallback_gpu.py).  The product library has none of it.  DMR_LIBRARY=<path> makes the package load another
build of the C ABI (that is how the tests reach the ablation build).

hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off is part of the product's
FP contract (see csrc/dmr_device.hpp), not a debugging flag.  -fno-slp-vectorize keeps the
compiler from pairing scalar f32 math into v_pk_* instructions: on gfx950 a packed op costs
~1.7x a scalar one and needs v_mov shuffles to form its operand pairs (scripts/micro/valu_rates.hip;
k_tri_forward 72 -> 63 us, k_tri_backward_hits 149 -> 134 us at C4 without it).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdmesh_renderer_hip.so")
LIB_ABLATION = os.path.join(HERE, "libdmesh_renderer_hip_ablation.so")
SOURCES = ["dmr_api.hip", "dmr_binning.hip", "dmr_tri.hip", "dmr_tet.hip"]
HEADERS = ["dmr_device.hpp", "dmr_kernels.hpp", os.path.join("..", "..", "include", "dmesh_renderer_amd.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, ablation: bool = False) -> str:
    lib = LIB_ABLATION if ablation else LIB
    if not force and not stale(lib):
        return lib
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wl,-rpath,/opt/rocm/lib", "-o", lib + ".tmp"]
    if ablation:
        cmd.append("-DDMR_ABLATION")
    cmd += os.environ.get("DMR_HIPCC_FLAGS", "").split()  # tuning experiments only
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))
