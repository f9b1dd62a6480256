"""sparse_utils.get_stats + rigl_amd/counting.py (the absent micronet_challenge `counting` dependency, restated) against
the reference's PUBLISHED numbers for its own ResNet-50 (/root/reference/README.md:32-44, 64-67): inference FLOPs relative
to dense and model size in MB.  The sizes agree to the README's three decimals, which pins the parameter / mask-bit /
bias accounting; the FLOP ratios to its two."""
import numpy as np
import pytest

from rigl_amd import counting, sparse_utils
from rigl_amd.workloads import shapes


def _mb(bits):
  return bits / 8 / 1e6


def test_resnet50_dense_matches_readme():
  flops, bits, s = sparse_utils.get_stats(shapes.resnet50_stat_layers(), 0.0, 'random')
  assert abs(flops / 8.2e9 - 1) < 0.005            # README.md:34  "8.2e9"
  assert round(_mb(bits), 3) == 102.122            # README.md:34
  assert s == 0.0


# (method, sparsity, first layer dense, README FLOPs ratio, README MB)   README.md:35-41, 64-67
README_ROWS = [
    ('erdos_renyi_kernel', 0.8, False, 0.42, 23.683),
    ('erdos_renyi_kernel', 0.9, False, 0.24, 13.499),
    ('random', 0.9, True, 0.13, 13.532),
    ('erdos_renyi_kernel', 0.95, False, 0.12, 8.399),
    ('random', 0.95, True, 0.08, 8.433),
    ('erdos_renyi_kernel', 0.99, True, 0.05, 4.354),
]


@pytest.mark.parametrize('method,sparsity,first_dense,ratio,mb', README_ROWS)
def test_resnet50_sparse_rows_match_readme(method, sparsity, first_dense, ratio, mb):
  layers = shapes.resnet50_stat_layers()
  dense = sparse_utils.get_stats(layers, 0.0, 'random')[0]
  custom = {'initial_conv': 0.0} if first_dense else {}
  flops, bits, real = sparse_utils.get_stats(layers, sparsity, method, custom_sparsities=custom)
  assert abs(flops / dense - ratio) < 0.006, flops / dense
  assert abs(_mb(bits) - mb) <= 0.0015, _mb(bits)
  assert abs(real - sparsity) < 1e-3


def test_mobilenet_dense_matches_readme():
  flops, _, _ = sparse_utils.get_stats(shapes.mobilenet_v1_stat_layers(), 0.0, 'random')
  assert abs(flops / 1.14e9 - 1) < 0.005           # README.md:50


def test_counting_rules():
  # a dense 3x3 conv, 8x8 input, stride 1: 64 outputs x 16 channels; dot products of length 27
  c = counting.Conv2D(8, [3, 3, 3, 16], (1, 1), 'same', True, 'relu')
  bits, mults, adds = counting.count_ops(c, 0.0, 32)
  n_out = 8 * 8 * 16
  assert bits == (3 * 3 * 3 * 16 + 16) * 32
  assert mults == 27 * n_out and adds == 26 * n_out + n_out + n_out
  # half of it pruned: the kept parameters, one mask bit per element, dot products half as long
  bits, mults, adds = counting.count_ops(c, 0.5, 32)
  assert bits == 3 * 3 * 3 * 16 * 32 * 0.5 + 3 * 3 * 3 * 16 + 16 * 32
  assert mults == 13.5 * n_out and adds == 12.5 * n_out + 2 * n_out
  # strides and 'same' padding: ceil((n - k + 1 + 2 (k // 2)) / s)
  assert counting.get_conv_output_size(224, 7, 'same', 2) == 112
  assert counting.get_conv_output_size(56, 3, 'same', 2) == 28
  assert counting.get_conv_output_size(56, 1, 'same', 2) == 28
  d = counting.DepthWiseConv2D(8, [3, 3, 16, 1], (2, 2), 'same', True, 'relu')
  bits, mults, adds = counting.count_ops(d, 0.0, 32)
  assert bits == (9 * 16 + 16) * 32 and mults == 9 * 4 * 4 * 16
  f = counting.FullyConnected([100, 10], True, 'relu')
  bits, mults, adds = counting.count_ops(f, 0.9, 32)
  assert np.isclose(mults, 100 * 0.1 * 10) and np.isclose(adds, (100 * 0.1 - 1) * 10 + 20)


def test_width_scaling_and_names():
  layers = [sparse_utils.StatLayer('conv2d', 'conv1', (3, 3, 3, 16), 32, (1, 1)),
            sparse_utils.StatLayer('conv2d', 'mid', (3, 3, 16, 16), 32, (1, 1)),
            sparse_utils.StatLayer('dense', 'conv_preds', (16, 10))]
  f1, b1, _ = sparse_utils.get_stats(layers, 0.0, 'random')
  f2, b2, _ = sparse_utils.get_stats(layers, 0.0, 'random', width=2.)
  # the first layer keeps its 3 inputs, the last its 10 outputs; everything else doubles
  n = 32 * 32
  want = (2 * 27 * 32 + 32) * n + (2 * 9 * 32 * 32 + 32) * n + 2 * 32 * 10 + 10
  assert f2 == want and f2 > f1 and b2 > b1
