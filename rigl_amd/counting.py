"""Operation / parameter counting for sparse networks, as `sparse_utils.get_stats` uses it
(/root/reference/rigl/sparse_utils.py:26, 436-447: `from google_research.micronet_challenge import counting`).

The module is a THIRD-PARTY dependency that is absent from /root/reference: `google_research/micronet_challenge/counting.py`
of the google-research monorepo (no pinned version: the reference imports it by path from the same checkout).  What is
restated here is its published algorithm (the MicroNet challenge's scoring rules) for the three layer types get_stats
builds -- Conv2D, DepthWiseConv2D, FullyConnected with padding 'same', bias and a ReLU -- and it is pinned to the
reference's own published numbers for its own models (README.md:32-67: ResNet-50 8.2e9 FLOPs dense, 0.42x / 0.23x /
0.24x / 0.13x / 0.05x at ERK-0.8 / uniform-0.8 / ERK-0.9 / uniform-0.9 / ERK-0.99, 102.122 / 23.683 MB; tests/test_get_stats.py).

Rules (multiplications and additions are counted separately and then summed by get_stats):
  * a sparse tensor of n elements at sparsity s stores n (1 - s) parameters of `param_bits` bits plus, if s > 0, a 1-bit
    mask per element;
  * a dot product of (effective) length L costs L multiplications and L - 1 additions; a sparse kernel's effective
    length is L (1 - s) -- "effective FLOPs", what the reference's 0.42x counts, as opposed to the dense-equivalent
    FLOPs the MFMA kernels of this package execute;
  * a bias is one parameter per output channel and one addition per output element; ReLU one addition-class
    operation (a comparison) per output element.
"""
import collections

import numpy as np

Conv2D = collections.namedtuple('Conv2D', ['input_size', 'kernel_shape', 'strides', 'padding', 'use_bias', 'activation'])
DepthWiseConv2D = collections.namedtuple('DepthWiseConv2D',
                                         ['input_size', 'kernel_shape', 'strides', 'padding', 'use_bias', 'activation'])
FullyConnected = collections.namedtuple('FullyConnected', ['kernel_shape', 'use_bias', 'activation'])


def get_conv_output_size(image_size, filter_size, padding, stride):
  """Output side of a square convolution: 'same' pads filter_size // 2 on each side."""
  if padding == 'same':
    pad = filter_size // 2
  elif padding == 'valid':
    pad = 0
  else:
    raise NotImplementedError('Padding: %s should be `same` or `valid`.' % padding)
  return int(np.ceil((image_size - filter_size + 1. + 2 * pad) / stride))


def get_sparse_size(tensor_shape, param_bits, sparsity):
  """Bits of a possibly sparse tensor: the kept parameters + a binary mask when anything is pruned."""
  n_elements = np.prod(tensor_shape)
  c_size = n_elements * param_bits * (1 - sparsity)
  if sparsity > 0:
    c_size += n_elements      # 1 bit per element
  return c_size


def get_flops_per_activation(activation):
  """(multiplications, additions) per output element."""
  if activation == 'relu':
    return 0, 1               # one comparison
  if activation is None:
    return 0, 0
  raise ValueError('activation %r is not one get_stats uses' % (activation,))


def count_ops(op, sparsity, param_bits):
  """(param_bits_total, n_multiplications, n_additions) of one layer at the given sparsity of its kernel."""
  flop_mults = flop_adds = param_count = 0
  if isinstance(op, Conv2D):
    k_size, k2, c_in, c_out = op.kernel_shape
    assert k_size == k2 and op.strides[0] == op.strides[1], 'square kernels and strides'
    param_count += get_sparse_size([k_size, k_size, c_in, c_out], param_bits, sparsity)
    vector_length = (k_size * k_size * c_in) * (1 - sparsity)
    n_out = get_conv_output_size(op.input_size, k_size, op.padding, op.strides[0])**2 * c_out
  elif isinstance(op, DepthWiseConv2D):
    k_size, k2, channels, mult = op.kernel_shape
    assert k_size == k2 and mult == 1 and op.strides[0] == op.strides[1]
    param_count += get_sparse_size([k_size, k_size, channels], param_bits, sparsity)
    vector_length = (k_size * k_size) * (1 - sparsity)
    n_out = get_conv_output_size(op.input_size, k_size, op.padding, op.strides[0])**2 * channels
    c_out = channels
  elif isinstance(op, FullyConnected):
    c_in, c_out = op.kernel_shape
    param_count += get_sparse_size([c_in, c_out], param_bits, sparsity)
    vector_length = c_in * (1 - sparsity)
    n_out = c_out
  else:
    raise ValueError('Encountered unknown operation %s.' % str(op))
  flop_mults += vector_length * n_out
  flop_adds += (vector_length - 1) * n_out
  if op.use_bias:
    param_count += c_out * param_bits
    flop_adds += n_out
  if op.activation:
    n_muls, n_adds = get_flops_per_activation(op.activation)
    flop_mults += n_muls * n_out
    flop_adds += n_adds * n_out
  return param_count, flop_mults, flop_adds
