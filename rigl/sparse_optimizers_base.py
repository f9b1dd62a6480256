"""rigl.sparse_optimizers_base -> rigl_amd.sparse_optimizers (HIP-backed)."""
from rigl_amd.sparse_optimizers import (  # noqa: F401
    SparseRigLOptimizerBase, SparseSETOptimizerBase, extract_number)
