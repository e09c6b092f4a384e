"""pysparselp_amd: the first-order sparse-LP hot path of PySparseLP on AMD MI355X.

``SparseLP.solve(method="admm" | "chambolle_pock_ppd" | "admm_blocks")`` with the reference's
signature (modules mirror the reference's: ``SparseLP``, ``ADMM``, ``ChambollePockPPD``, ``ADMMBlocks``,
``gaussSiedel``, ``MPSparser``, ``netlib``, ``tools``; ``device`` / ``scale`` / ``problems`` / ``parallel`` hold the
device-resident, at-scale and multi-GPU entry points); the inner loops run in hand-written HIP kernels (libslp_hip.so,
C ABI in include/slp_hip.h).  There is no CPU fallback: without the built
library and a HIP device every solver call raises ``SlpError``.
"""
from ._lib import ORDER_AUTO, ORDER_SEQUENTIAL, ORDER_TREE, SlpError  # noqa: F401

__all__ = ["ORDER_AUTO", "ORDER_SEQUENTIAL", "ORDER_TREE", "SlpError"]
