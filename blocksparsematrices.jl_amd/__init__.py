"""MI355X-native block-sparse mat-vec engine: drop-in for the mul! hot path of
djukic14/BlockSparseMatrices.jl (BlockSparseMatrix / SymmetricBlockMatrix / VBCRS x vector).

The directory name contains a dot, so import it through the root-level shim:  `import bsm_amd`.
"""
from . import _lib  # noqa: F401
from .matrices import *  # noqa: F401,F403
from .matrices import __all__ as _m_all
from . import synthetic  # noqa: F401

__all__ = list(_m_all) + ["synthetic"]
