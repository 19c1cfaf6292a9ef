"""The C-ABI library loads without a GPU and exports every symbol include/dmesh_renderer_amd.h declares;
the Python binding fails loudly instead of falling back when it cannot run."""
import ctypes
import os
import re

import pytest
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dmesh_renderer_amd.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(dmr_[a-z_0-9]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_header_declares_the_four_entry_points():
    fns = _declared_functions()
    for n in ("dmr_tri_forward", "dmr_tri_backward", "dmr_tet_forward", "dmr_tet_backward", "dmr_last_error"):
        assert n in fns


def test_library_exports_every_declared_symbol():
    import capi_ctypes as _lib
    from dmesh_renderer_amd import build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in _declared_functions():
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # and the binding table covers exactly the header
    assert sorted(_lib.EXPORTS) == _declared_functions()


def test_binding_loads_and_reports_arch():
    import capi_ctypes as _lib
    lib = _lib.load()
    assert lib.dmr_build_arch() == b"gfx950"
    assert lib.dmr_abi_version() == _lib.ABI_VERSION
    assert lib.dmr_stage_name(6) == b"k_tri_backward_pix" and lib.dmr_stage_name(11) == b"k_tri_backward_hits"


def test_scene_struct_layout_matches_header():
    """dmr_scene: 6 int32, 14 pointers, 5 int32 (natural alignment)."""
    import capi_ctypes as _lib
    assert ctypes.sizeof(_lib.Scene) == 6 * 4 + 14 * 8 + 5 * 4 + 4  # + tail padding to 8
    assert _lib.Scene.background.offset == 24 and _lib.Scene.ray_random_seed.offset == 24 + 14 * 8


def test_compiled_binding_is_the_only_one():
    """`_C` is a compiled extension module over the C ABI (no ctypes / pure-Python binding left in the product) and
    it is bound to the in-tree HIP library."""
    import dmesh_renderer_amd as dmr
    from dmesh_renderer_amd import _C
    assert _C.__file__.endswith(".so") and os.path.dirname(_C.__file__) == os.path.dirname(dmr.__file__)
    assert _C.library_path() == os.path.join(os.path.dirname(dmr.__file__), "libdmesh_renderer_hip.so")
    assert _C.ABI_VERSION == 4 and _C.build_arch() == "gfx950" and _C.NUM_STAGES == 12
    assert _C.stage_name(_C.STAGE_TRI_BACKWARD_HITS) == "k_tri_backward_hits"
    for n in ("render_tris", "render_tris_backward", "render_tets", "render_tets_backward"):  # ext.cpp:6-11
        assert callable(getattr(_C, n))
    pkg = os.path.dirname(dmr.__file__)
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "ctypes" not in open(os.path.join(pkg, f)).read(), f


def test_no_cpu_fallback():
    """CPU tensors are refused: the product path has no CPU implementation."""
    from dmesh_renderer_amd import _C, scenes
    d = scenes.layered_sheets(1, 3, 1, 32, 32)
    args = scenes.c_args(d)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.render_tris(*args, 32, 32)
    dt = scenes.kuhn_tets(2, 1, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.render_tets(*scenes.c_args(dt, tet=True), 32, 32, 0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dmesh_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "dmr_oracle" not in txt, f


def test_product_library_has_no_ablation_switch():
    """The DMR_ABLATE switches (timing ablations, forced fallback) are compiled out of the product library: no getenv
    import at all; the ablation build (build.py --ablation) is a separate file that has it."""
    import subprocess
    from dmesh_renderer_amd import build
    prod, abl = build.build(), build.build(ablation=True)
    syms = lambda lib: subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms(prod)
    assert "getenv" in syms(abl)
    for f in ("dmr_tri.hip", "dmr_tet.hip", "dmr_binning.hip", "dmr_api.hip"):
        src = open(os.path.join(ROOT, "dmesh_renderer_amd", "csrc", f)).read()
        for m in re.finditer(r"getenv", src):
            before = src[:m.start()]
            assert before.rfind("#ifdef DMR_ABLATION") > before.rfind("#endif"), f"{f}: getenv outside DMR_ABLATION"
