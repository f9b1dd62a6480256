"""K1 parity at the kernel-selection settings that matter, through tests/k1_check.py (a subprocess per setting: the
library reads RIGL_* once per process).  The checker is tests/convref.py's fp64 convolution of the same bf16
operands; bound per element 2^-8 |ref| + 1e-5 sum|a||b| (fwd, dgrad), 1e-5 sum|a||b| (wgrad) -- the north star's
fp32 tolerance relative to the magnitude of the dot product."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=1500):
  e = dict(os.environ)
  e.update(env or {})
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'k1_check.py')] + args, env=e,
                     capture_output=True, text=True, timeout=timeout)
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert lines, 'no verdict line; stderr tail: %s' % r.stderr[-3000:]
  out = json.loads(lines[-1])
  assert r.returncode == 0 and out['ok'], '%s\n%s' % (out, r.stdout[-3000:])
  return out


PP_SETTINGS = [
    {},                                                                          # the built-in selection rules
    {'RIGL_PP_FWD': '1', 'RIGL_PP_DGRAD': '1'},                                  # 256x256 wherever legal (two phases per K-tile)
    {'RIGL_PP_FWD': '2', 'RIGL_PP_DGRAD': '2'},                                  # 128x256 (one phase, three stages)
    {'RIGL_PP_FWD': '3', 'RIGL_PP_DGRAD': '3'},                                  # 256x128
    {'RIGL_PP_FWD': '4', 'RIGL_PP_DGRAD': '4'},                                  # 512x128
    {'RIGL_PP_FWD': '0', 'RIGL_PP_DGRAD': '0'},                                  # the igemm body everywhere
    {'RIGL_PP_KSPLIT': '0'},                                                     # no K split of the few-tile forwards (on by default)
]


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', PP_SETTINGS)
def test_small_shapes(env):
  out = _run(['--set', 'small'], env)
  assert out['cases'] == 12


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', PP_SETTINGS[:5])
def test_pingpong_edge_shapes(env):
  """1 .. 36 K-tiles (every prologue / tail branch of the ring), ragged row tiles, stride-2 forwards, TF-SAME asymmetric
  padding, 128- / 256-wide column tiles -- each ping-pong tile forced wherever it is legal, against the fp64 reference."""
  out = _run(['--set', 'pp'], env)
  assert out['cases'] == 12


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', PP_SETTINGS[:3] + PP_SETTINGS[4:5] + [
    {'RIGL_BWD1X1': '0', 'RIGL_STEM_DIRECT': '0', 'RIGL_WGRAD_IL': '0', 'RIGL_C3X3': '0', 'RIGL_ROWSTREAM': '0', 'RIGL_BWDSLICE': '0'},   # the generic bodies on the layers with kernels of their own
])
def test_resnet50_layer_shapes_at_batch_128(env):
  """All distinct ResNet-50 conv shapes at the benchmarked per-GPU batch: fwd, fwd + statistics,
  dgrad, dgrad + addend, wgrad and the one-call backward, under the default kernel selection (the ping-pong body on the
  long-reduction forwards) and with each ping-pong tile forced wherever legal, forward and dgrad -- so every kernel that
  can run in the step is pinned directly against the fp64 reference at the benchmarked sizes."""
  out = _run(['--set', 'resnet50', '--batch', '128'], env, timeout=3000)
  assert out['cases'] == 23


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [{}, {'RIGL_STEM_DIRECT': '0'}])
def test_imagenet_stem_shapes(env):
  """The 7x7 / 2 stem on the LDS-resident-patch kernels (stem.hpp) and, with the knob off, on the generic bodies over the
  padded copy: image borders on all sides, H != W, TF-SAME top padding, the statistics parts of 16 x 16 tiles."""
  out = _run(['--set', 'stem'], env)
  assert out['cases'] == 6


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [{}, {'RIGL_C3X3': '0'}])
def test_slab_resident_3x3_shapes(env):
  """The 3x3 / 64 -> 64 kernels with the input patch resident in LDS and the filter in registers (c3x3.hpp: forward, dgrad
  (+ addend), weight gradient, statistics per tile) on whole and partial tiles, odd and maximal widths; with the knob off
  the same shapes on the generic bodies (pins the cases themselves)."""
  out = _run(['--set', 'c3'], env)
  assert out['cases'] == 6


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [{'RIGL_ROWSTREAM': '2'}, {}, {'RIGL_ROWSTREAM': '0'}])
def test_row_streaming_1x1_shapes(env):
  """The row-streaming 1x1 body (rowstream.hpp: activation / gradient rows global -> registers, the filter slice stationary
  in LDS, barrier-free waves; forward with statistics, dgrad with and without addend) in every (reduction, slice width)
  instantiation, 1 .. 32 column slices, fragments of fewer than 32 rows, ragged last fragments -- forced onto every legal
  shape, then under the default selection, then with the body off (pins the cases themselves)."""
  out = _run(['--set', 'rs'], env)
  assert out['cases'] == 13


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [{}, {'RIGL_BWDSLICE': '0'}])
def test_channel_sliced_single_pass_backward_shapes(env):
  """The channel-sliced single-pass backward of the many-input-channel 1x1 layers (bwdslice.hpp: W slice in registers, dY
  and the X slice through one LDS-DMA ring, dX and the slice's dW from the same tiles) on 1 .. 32 slices, cout 128 / 256
  and 512 (slices of 64 channels: k_bwdslice64), ragged last tiles: dgrad (+ addend), the one-call backward (dX bit-equal to dgrad + addend, dW deterministic) against
  the fp64 reference; with the knob off the same shapes on the shared-launch bodies (pins the cases themselves)."""
  out = _run(['--set', 'bs'], env)
  assert out['cases'] == 12
