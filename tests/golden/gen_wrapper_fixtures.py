"""Generates tests/golden/wrapper_{tri,tet}.npz.  Run in the build container only:

    python tests/golden/gen_wrapper_fixtures.py

The REFERENCE Python wrapper (/root/reference/dmesh_renderer/__init__.py, imported read-only) is run on
seeded synthetic scenes with `dmesh_renderer._C` provided by tests/oracle_C.py (the CPU oracle), because
the reference's CUDA extension cannot be built here.  The fixtures therefore pin the Python boundary
(transposes, inverses, dtype casts, gradient routing) and the oracle's numbers on these scenes; they are
data (inputs are regenerated from the seeds in tests/test_wrapper_cpu.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import test_wrapper_cpu as T  # noqa: E402


def main():
    ref = T._reference_module()
    c, d, g = T._tri_run(ref, T._tri_scene(), T.TRI["H"], T.TRI["W"])
    np.savez_compressed(os.path.join(HERE, "wrapper_tri.npz"), color=c.numpy(), depth=d.numpy(),
                        **{"grad_" + k: v.numpy() for k, v in g.items()})
    c, d, a, g = T._tet_run(ref, T._tet_scene(), T.TET["H"], T.TET["W"])
    np.savez_compressed(os.path.join(HERE, "wrapper_tet.npz"), color=c.numpy(), depth=d.numpy(), active=a.numpy(),
                        **{"grad_" + k: v.numpy() for k, v in g.items()})
    print("wrote fixtures")


if __name__ == "__main__":
    main()
