"""ctypes binding of the C ABI in ``include/rigl_hip.h`` (librigl_hip.so).

There is NO fallback: if the library is missing or a call fails, this module
raises.  The product never computes the hot path any other way.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2   # include/rigl_hip.h RIGL_ABI_VERSION this mirror was written against
LIB_PATH = os.environ.get('RIGL_HIP_LIB') or os.path.join(_HERE, 'lib', 'librigl_hip.so')   # override: development builds

RIGL_OK = 0
RIGL_EINVAL = -1
RIGL_ELAUNCH = -2
RIGL_EWORKSPACE = -3
RIGL_EUNSUPPORTED = -4
COUNTS_PER_LAYER = 8
PROF_KINDS = ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'prune_regrow',
              'sgd_momentum', 'pack_weights', 'conv_bwd', 'depthwise')

GROW_ZEROS, GROW_GRAD_SCALE, GROW_GRAD_SIGN, GROW_EXPLICIT = 0, 1, 2, 3
MOMRESET_ZEROS, MOMRESET_GRAD = 0, 1


class RiglError(RuntimeError):

  def __init__(self, code, msg):
    super().__init__('librigl_hip: error %d: %s' % (code, msg))
    self.code = code


class PruneRegrowLayer(C.Structure):
  _fields_ = [('n', C.c_int64), ('w', C.c_void_p), ('momentum', C.c_void_p),
              ('mask_bits', C.c_void_p), ('dense_grad', C.c_void_p),
              ('drop_noise', C.c_void_p), ('score_drop', C.c_void_p),
              ('score_grow', C.c_void_p), ('grow_values', C.c_void_p)]


class PruneRegrowParams(C.Structure):
  _fields_ = [('drop_fraction', C.c_float), ('grow_init_mode', C.c_int32),
              ('grow_init_div', C.c_float), ('momentum_reset_mode', C.c_int32),
              ('initial_acc_scale', C.c_float), ('reinit_when_same', C.c_int32)]


class TopkLayer(C.Structure):
  _fields_ = [('score', C.c_void_p), ('n', C.c_int64), ('n_keep', C.c_int64),
              ('mask_bits', C.c_void_p)]


class PackLayer(C.Structure):
  _fields_ = [('w', C.c_void_p), ('mask_bits', C.c_void_p),
              ('hwio', C.c_void_p), ('ohwi', C.c_void_p), ('k', C.c_int32),
              ('cout', C.c_int32)]


class ConvDesc(C.Structure):
  _fields_ = [('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
              ('cin', C.c_int32), ('ho', C.c_int32), ('wo', C.c_int32),
              ('cout', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32),
              ('stride_h', C.c_int32), ('stride_w', C.c_int32),
              ('pad_top', C.c_int32), ('pad_left', C.c_int32)]


class BnReduceFuse(C.Structure):
  """RiglBnReduceFuse: the batch norm whose backward reductions ride in a dgrad epilogue."""
  _fields_ = [('x', C.c_void_p), ('relu_bits', C.c_void_p), ('params', C.c_void_p), ('relu', C.c_int32),
              ('partial', C.c_void_p), ('partial_floats', C.c_size_t)]


class RandomItem(C.Structure):
  """RiglRandomItem"""
  _fields_ = [('out', C.c_void_p), ('n', C.c_int64), ('seed0', C.c_int32), ('seed1', C.c_int32),
              ('dist', C.c_int32), ('scale', C.c_float), ('shift', C.c_float)]


class ProfLaunch(C.Structure):
  """RiglProfLaunch: one timed K1 / K2 / K3 dispatch."""
  _fields_ = [('kind', C.c_int32), ('tag', C.c_int32 * 6), ('ms', C.c_float)]


# name -> (restype, argtypes); every symbol declared in include/rigl_hip.h
_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    'rigl_version': (C.c_int, []),
    'rigl_last_error': (C.c_char_p, []),
    'rigl_mask_pack': (C.c_int, [_P, _P, _I64, _P]),
    'rigl_mask_unpack': (C.c_int, [_P, _P, _I64, _P]),
    'rigl_prune_regrow_workspace_bytes': (_SZ, [C.POINTER(_I64), _I32]),
    'rigl_prune_regrow': (C.c_int, [C.POINTER(PruneRegrowLayer), _I32,
                                    C.POINTER(PruneRegrowParams), _P, _P, _SZ,
                                    _P]),
    'rigl_prune_regrow_selections_workspace_bytes': (_SZ, [_I64]),
    'rigl_prune_regrow_selections': (C.c_int, [C.POINTER(PruneRegrowLayer), C.POINTER(PruneRegrowParams), _P, _P, _P, _P, _P,
                                               _P, _SZ, _P]),
    'rigl_topk_mask': (C.c_int, [_P, _I64, _I64, _P, _P, _SZ, _P]),
    'rigl_topk_mask_batched': (C.c_int, [C.POINTER(TopkLayer), _I32, _P, _SZ, _P]),
    'rigl_masked_sgd_momentum': (C.c_int, [_I64, _P, _P, _P, _P, _F, _F, _F, _F,
                                           _I32, _P, _P]),
    'rigl_pack_weights': (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P]),
    'rigl_pack_weights_batched': (C.c_int, [C.POINTER(PackLayer), _I32, _P]),
    'rigl_conv2d_workspace_bytes': (_SZ, [C.POINTER(ConvDesc), _I32]),
    'rigl_masked_conv2d_fwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P,
                                         _SZ, _P]),
    'rigl_masked_conv2d_dgrad': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P,
                                           _SZ, _P]),
    'rigl_conv2d_stats_parts': (_I32, [C.POINTER(ConvDesc)]),
    'rigl_masked_conv2d_fwd_stats': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P,
                                               _SZ, _P, _SZ, _P]),
    'rigl_conv2d_fwd_takes_bn_input': (_I32, [C.POINTER(ConvDesc)]),
    'rigl_masked_conv2d_fwd_bnrelu': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    'rigl_masked_conv2d_dgrad_acc': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P,
                                               _P, _P, _SZ, _P]),
    'rigl_conv2d_dgrad_stats_parts': (_I32, [C.POINTER(ConvDesc)]),
    'rigl_masked_conv2d_bwd_bn': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _SZ,
                                            C.POINTER(BnReduceFuse), _P]),
    'rigl_masked_conv2d_bwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_masked_conv2d_bwd_masked': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_conv2d_bwd_takes_masked_addend': (_I32, [C.POINTER(ConvDesc)]),
    'rigl_masked_conv2d_bwd_grid': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_masked_conv2d_bwd_sub': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I32, _I32, _P, _P, _P, _SZ, _P]),
    'rigl_masked_conv2d_wgrad': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P,
                                           _SZ, _P]),
    'rigl_conv2d_fwd_ref': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_conv2d_dgrad_ref': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_conv2d_wgrad_ref': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_masked_conv2d_fwd_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    'rigl_masked_conv2d_dgrad_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    'rigl_conv2d_wgrad_f32_workspace_bytes': (_SZ, [C.POINTER(ConvDesc)]),
    'rigl_masked_conv2d_wgrad_f32': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _SZ, _P]),
    'rigl_depthwise_conv2d_workspace_bytes': (_SZ, [C.POINTER(ConvDesc)]),
    'rigl_depthwise_conv2d_fwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_depthwise_conv2d_stats_parts': (_I32, [C.POINTER(ConvDesc)]),
    'rigl_depthwise_conv2d_fwd_stats': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _SZ, _P]),
    'rigl_depthwise_conv2d_dgrad': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_depthwise_conv2d_wgrad': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _SZ, _P]),
    'rigl_bn_workspace_bytes': (_SZ, [_I64, _I32]),
    'rigl_bn_fwd': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _F, _F, _I32, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_bn_fwd_stats': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _F, _F, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P]),
    'rigl_bn_bwd': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_bn_bwd_stats': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _P, _SZ, _P]),
    'rigl_crc32c': (C.c_uint32, [_P, _SZ, C.c_uint32]),
    'rigl_stateless_random': (C.c_int, [_P, _I64, _I32, _I32, _I32, _F, _F, _P]),
    'rigl_stateless_random_batched': (C.c_int, [C.POINTER(RandomItem), _I32, _P]),
    'rigl_maxpool_fwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_maxpool_bwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    'rigl_bn_fwd_statistics': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I32, _P, _SZ, _P]),
    'rigl_bn_add_bn_fwd': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P]),
    'rigl_bn_add_bn_bwd_workspace_bytes': (_SZ, [_I64, _I32]),
    'rigl_bn_add_bn_bwd': (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_bn_relu_maxpool_fwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    'rigl_bn_relu_maxpool_bwd_workspace_bytes': (_SZ, [C.POINTER(ConvDesc)]),
    'rigl_bn_relu_maxpool_bwd': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'rigl_global_avgpool_fwd': (C.c_int, [_I32, _I32, _I32, _P, _P, _P]),
    'rigl_global_avgpool_bwd': (C.c_int, [_I32, _I32, _I32, _P, _P, _P]),
    'rigl_softmax_xent': (C.c_int, [_I32, _I32, _P, _P, _F, _F, _P, _P, _P]),
    'rigl_prof_enable': (C.c_int, [_I32]),
    'rigl_prof_collect': (C.c_int, [C.POINTER(C.c_double), C.POINTER(_I64)]),
    'rigl_prof_collect_launches': (C.c_int, [C.POINTER(ProfLaunch), _I64, C.POINTER(_I64)]),
    'rigl_probe_mfma_bf16': (C.c_int, [_I32, _I32, _P, _P]),
    'rigl_tune_set': (C.c_int, [C.c_char_p, _I32]),
    'rigl_tune_get': (_I32, [C.c_char_p, _I32]),
    'rigl_tune_generation': (C.c_uint64, []),
    'rigl_tune_generation_addr': (C.c_void_p, []),
    'rigl_tune_unset': (C.c_int, [C.c_char_p]),
}

_lib = None


def load():
  """Loads librigl_hip.so (after torch, so both share one HIP runtime)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RiglError(
        RIGL_ELAUNCH, 'HIP extension not built: %s is missing; run '
        '`python -c "import __graft_entry__ as g; g.build()"` or '
        '`make -C rigl_amd/csrc`' % LIB_PATH)
  try:
    import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  except ImportError:
    pass
  lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
    fn.restype = res
    fn.argtypes = args
  if lib.rigl_version() != ABI_VERSION:
    raise RiglError(RIGL_EINVAL, 'ABI version mismatch')
  _lib = lib
  return lib


def check(rc):
  if rc != RIGL_OK:
    raise RiglError(rc, load().rigl_last_error().decode('utf-8', 'replace'))
