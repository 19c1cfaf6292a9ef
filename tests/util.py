"""Shared helpers for the parity tests (thin re-exports so tests read `from util import ...`)."""
from dmesh_renderer_amd.scenes import SUM_ORDER_TOL, sum_order_tol, c_args, elementwise_close, max_abs_err, rel_err, upstream_grads  # noqa: F401
