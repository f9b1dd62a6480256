"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol that include/rigl_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  txt = open(os.path.join(ROOT, 'include', 'rigl_hip.h')).read()
  txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
  return sorted(set(re.findall(r'\b(rigl_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_the_hot_path():
  syms = _declared_symbols()
  for must in ['rigl_prune_regrow', 'rigl_masked_sgd_momentum',
               'rigl_masked_conv2d_fwd', 'rigl_masked_conv2d_dgrad',
               'rigl_masked_conv2d_wgrad', 'rigl_mask_pack',
               'rigl_mask_unpack', 'rigl_last_error', 'rigl_version']:
    assert must in syms


def test_library_exports_every_declared_symbol():
  from rigl_amd import _lib
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = _lib.load()
  for s in _declared_symbols():
    assert hasattr(lib, s), 'librigl_hip.so does not export %s' % s
    assert s in _lib.SIGNATURES, 'no ctypes signature for %s' % s
  assert lib.rigl_version() == 2
  assert lib.rigl_last_error() == b''


def test_argument_errors_are_reported_not_thrown():
  """Bad arguments return RIGL_EINVAL + message; nothing touches a device."""
  from rigl_amd import _lib
  lib = _lib.load()
  assert lib.rigl_mask_pack(None, None, 10, None) == _lib.RIGL_EINVAL
  assert b'rigl_mask_pack' in lib.rigl_last_error()
  assert lib.rigl_prune_regrow(None, 1, None, None, None, 0, None) == _lib.RIGL_EINVAL
  d = _lib.ConvDesc(1, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1, 1, 1)
  assert lib.rigl_masked_conv2d_fwd(ctypes.byref(d), None, None, None, None, 0, None) == _lib.RIGL_EINVAL
  d.cout = 10
  assert lib.rigl_masked_conv2d_fwd(ctypes.byref(d), 1, 1, 1, None, 0, None) == _lib.RIGL_EUNSUPPORTED
  # the round-4 entry points: fp32 twin, subsampled addend, grid backward
  d.cout = 8
  assert lib.rigl_masked_conv2d_fwd_f32(ctypes.byref(d), None, None, None, None, None) == _lib.RIGL_EINVAL
  assert b'rigl_masked_conv2d_fwd_f32' in lib.rigl_last_error()
  assert lib.rigl_masked_conv2d_dgrad_f32(ctypes.byref(d), None, None, None, None, None, None) == _lib.RIGL_EINVAL
  assert lib.rigl_masked_conv2d_wgrad_f32(ctypes.byref(d), None, None, None, None, 0, None) == _lib.RIGL_EINVAL
  assert lib.rigl_conv2d_wgrad_f32_workspace_bytes(None) == 0
  assert lib.rigl_masked_conv2d_bwd_sub(ctypes.byref(d), 1, 1, 1, 1, 0, 2, 1, 1, None, 0, None) == _lib.RIGL_EINVAL
  assert b'subsampling' in lib.rigl_last_error()
  # ... the grid backward is for strided 1x1 convs without padding only
  assert lib.rigl_masked_conv2d_bwd_grid(ctypes.byref(d), 1, 1, 1, 1, 1, None, 0, None) == _lib.RIGL_EUNSUPPORTED
  g = _lib.ConvDesc(1, 8, 8, 8, 4, 4, 8, 1, 1, 2, 2, 0, 0)
  assert lib.rigl_masked_conv2d_bwd_grid(ctypes.byref(g), None, None, None, None, None, None, 0, None) == _lib.RIGL_EINVAL
  assert lib.rigl_tune_unset(None) == _lib.RIGL_EINVAL


def test_ops_refuse_cpu_tensors():
  """The product has no CPU path: handing it CPU tensors fails loudly."""
  import torch
  from rigl_amd import ops
  with pytest.raises(Exception) as ei:
    ops.mask_pack(torch.ones(8))
  assert 'GPU' in str(ei.value)
