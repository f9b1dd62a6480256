"""rigl.sparse_optimizers -> rigl_amd.sparse_optimizers (HIP-backed)."""
from rigl_amd.sparse_optimizers import (  # noqa: F401
    PruningGetterMixin, SparseDNWOptimizer, SparseMomentumOptimizer,
    SparseRigLOptimizer, SparseSnipOptimizer,
    SparseSETOptimizer, SparseStaticOptimizer, extract_number, get_grow_grads)
from rigl_amd.sparse_optimizers import PruningGetterMixin as PruningGetterTf1Mixin  # noqa: F401
