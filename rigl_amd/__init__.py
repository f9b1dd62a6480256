"""rigl_amd -- MI355X (gfx950) native RigL hot path.

Hand-written HIP kernels behind a C ABI (``include/rigl_hip.h``), hosted from
Python with the reference's own API surface:

  rigl_amd.sparse_optimizers   SparseRigLOptimizer, SparseSETOptimizer, ...
  rigl_amd.sparse_utils        get_mask_random, get_sparsities, get_mask_init_fn
  rigl_amd.pruning_layers      sparse_conv2d / sparse_fully_connected layers
  rigl_amd.ops                 tensor-level wrappers over the C ABI

(aliased under the reference's module names in the top-level ``rigl`` package).
"""
__version__ = '0.1.0'
