"""rigl.sparse_utils -> rigl_amd.sparse_utils."""
from rigl_amd.sparse_utils import (  # noqa: F401
    DEFAULT_ERK_SCALE, calculate_sparsity, get_mask_init_fn, get_mask_random,
    get_mask_random_numpy, get_n_zeros, get_sparsities,
    get_sparsities_erdos_renyi, get_sparsities_uniform, get_stats,
    mask_extract_name_fn)
