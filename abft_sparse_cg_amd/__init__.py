"""abft_sparse_cg_amd -- MI355X-native `hip` target for abft-sparse-cg's CGContext
plugin surface: CSR/COO SpMV with the reference's per-element software ECC fused
into the load path, dot / calc_xr / calc_p, all hand-written HIP for gfx950
behind the C ABI of include/abft_hip.h (libabft_hip.so).

Importing the package loads the shared library and fails if it is missing:
there is no CPU implementation in here.
"""
from . import capi
from .capi import MODES, AbftError  # noqa: F401

capi.load()

from .context import FatalEvent, HIPContext, cg_solve  # noqa: E402,F401
