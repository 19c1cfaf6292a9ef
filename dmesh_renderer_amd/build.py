"""Builds the two native pieces of the package, in-tree:

  * libdmesh_renderer_hip.so -- the C ABI (include/dmesh_renderer_amd.h) and every gfx950 kernel, with hipcc;
  * _C.<python ext suffix>   -- `dmesh_renderer_amd._C`, the compiled PyTorch-ROCm binding over that C ABI
                                (csrc/dmr_torch.cpp: pybind11 + ATen, no kernels), with g++.

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
HEADERS = ["dmr_device.hpp", "dmr_kernels.hpp", "dmr_sort.hpp", os.path.join("..", "..", "include", "dmesh_renderer_amd.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _glue_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_C" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


GLUE_SOURCES = ["dmr_torch.cpp", os.path.join("..", "..", "include", "dmesh_renderer_amd.h")]


def glue_stale() -> bool:
    out = _glue_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in GLUE_SOURCES) or os.path.getmtime(os.path.abspath(__file__)) > t


def build_glue(force: bool = False, verbose: bool = False) -> str:
    """dmesh_renderer_amd._C: host C++ only (shape checks, allocation through PyTorch's caching allocator, the current
    HIP stream, raw-pointer hand-off to the C ABI, which it loads with dlopen)."""
    out = _glue_path()
    if not force and not glue_stale():
        return out
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("g++ not found: the PyTorch binding cannot be built")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    cmd += [os.path.join(CSRC, "dmr_torch.cpp"), "-o", out + ".tmp", "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python", "-Wl,-rpath," + tlib, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def build_all(force: bool = False, verbose: bool = False):
    """Everything the package needs at import: the HIP library and the binding (nothing if both are up to date)."""
    res = build(force=force, verbose=verbose), build_glue(force=force, verbose=verbose)
    pkg = sys.modules.get(__package__ or "dmesh_renderer_amd")
    if pkg is not None and type(getattr(pkg, "_C", None)).__name__ == "_NotBuilt":  # imported before it was built
        import importlib
        pkg._C = importlib.import_module(pkg.__name__ + "._C")
    return res


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
    if "--ablation" not in sys.argv:
        print(build_glue(force="--force" in sys.argv, verbose=True))
