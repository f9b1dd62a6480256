"""Thin tensor-level wrappers over the C ABI (``include/rigl_hip.h``).

PyTorch is used for device memory and streams only: every function here takes
CUDA(=HIP) tensors, checks dtype/contiguity, and enqueues the hand-written
gfx950 kernels on ``torch.cuda.current_stream()``.  No function has a CPU or
PyTorch-op fallback; calling one without a GPU / without the built library
raises ``RiglError``.
"""
import ctypes as C
import os

import torch

from rigl_amd import _lib
from rigl_amd._lib import (ConvDesc, PackLayer, PruneRegrowLayer,
                           PruneRegrowParams, RiglError, check)

_workspaces = {}


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
  """hipStream_t of torch's current stream on the current device (the raw-handle
  query is ~10x cheaper than building a torch.cuda.Stream object per launch)."""
  if _raw_stream is not None and _cur_device is not None:
    return C.c_void_p(_raw_stream(_cur_device()))
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
  return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req(t, dtype, name, allow_none=False):
  if t is None:
    if allow_none:
      return
    raise ValueError('%s is None' % name)
  if not t.is_cuda:
    raise RiglError(_lib.RIGL_EINVAL,
                    '%s must live on the GPU (no CPU path exists)' % name)
  if t.dtype != dtype:
    raise TypeError('%s: expected %s, got %s' % (name, dtype, t.dtype))
  if not t.is_contiguous():
    raise ValueError('%s must be contiguous' % name)


# Bumped whenever a scratch buffer is (re)allocated: a captured HIP graph holds the OLD buffer's address, so
# train.GraphedStep drops its graphs when the generation it captured under is no longer current.
WORKSPACE_GENERATION = 0


def workspace(nbytes, device, tag=''):
  """Grow-only scratch buffer per device and user (stream-ordered reuse)."""
  global WORKSPACE_GENERATION
  key = (torch.device(device).index or 0, tag)
  ws = _workspaces.get(key)
  if ws is None or ws.numel() < nbytes:
    # grow geometrically (at most O(log) reallocations per user); the outgrown buffer is simply released -- kernels already
    # enqueued on it are safe (the caching allocator is stream-ordered) and train.GraphedStep compares WORKSPACE_GENERATION
    # before EVERY replay, so a captured graph never runs against a released buffer
    grown = max(int(nbytes), 1 << 20, (ws.numel() * 3 // 2) if ws is not None else 0)
    ws = torch.empty(grown, dtype=torch.uint8, device=device)
    _workspaces[key] = ws
    WORKSPACE_GENERATION += 1
  return ws


def n_mask_words(n):
  return (int(n) + 31) // 32


# ----------------------------------------------------------------------------
# mask bitmap
# ----------------------------------------------------------------------------
def mask_pack(mask01, out=None):
  """float 0/1 mask (any shape) -> int32 bitmap words (flat C order)."""
  _req(mask01, torch.float32, 'mask01')
  n = mask01.numel()
  if out is None:
    out = torch.zeros(n_mask_words(n), dtype=torch.int32, device=mask01.device)
  _req(out, torch.int32, 'bits')
  check(_lib.load().rigl_mask_pack(_ptr(mask01), _ptr(out), n, _stream()))
  return out


def mask_unpack(bits, shape):
  _req(bits, torch.int32, 'bits')
  out = torch.empty(shape, dtype=torch.float32, device=bits.device)
  check(_lib.load().rigl_mask_unpack(_ptr(bits), _ptr(out), out.numel(),
                                     _stream()))
  return out


# ----------------------------------------------------------------------------
# K2
# ----------------------------------------------------------------------------
def prune_regrow(layers, drop_fraction, grow_init_mode=_lib.GROW_ZEROS,
                 grow_init_div=1.0, momentum_reset_mode=_lib.MOMRESET_GRAD,
                 initial_acc_scale=0.0, reinit_when_same=False):
  """Runs the fused prune/regrow update on a list of layers, in place.

  Each layer is a dict with tensors: ``w`` (fp32), ``mask_bits`` (int32),
  optional ``momentum``, ``dense_grad``, ``drop_noise``, ``score_drop``,
  ``score_grow``, ``grow_values`` (all fp32, same numel as ``w``).
  Returns an int32 tensor [n_layers, 8] of counts (see rigl_hip.h).
  """
  lib = _lib.load()
  nl = len(layers)
  if nl == 0:
    return torch.zeros((0, _lib.COUNTS_PER_LAYER), dtype=torch.int32)
  arr = (PruneRegrowLayer * nl)()
  ns = (C.c_int64 * nl)()
  dev = None
  for i, l in enumerate(layers):
    w = l.get('w')
    ref = w if w is not None else l['score_drop']
    n = ref.numel()
    dev = ref.device
    for key in ('w', 'momentum', 'dense_grad', 'drop_noise', 'score_drop',
                'score_grow', 'grow_values'):
      t = l.get(key)
      _req(t, torch.float32, key, allow_none=True)
      if t is not None and t.numel() != n:
        raise ValueError('layer %d: %s has %d elements, expected %d' %
                         (i, key, t.numel(), n))
    _req(l['mask_bits'], torch.int32, 'mask_bits')
    if l['mask_bits'].numel() < n_mask_words(n):
      raise ValueError('layer %d: mask_bits too small' % i)
    arr[i].n = n
    arr[i].w = w.data_ptr() if w is not None else None
    for key in ('momentum', 'dense_grad', 'drop_noise', 'score_drop',
                'score_grow', 'grow_values'):
      t = l.get(key)
      setattr(arr[i], key, t.data_ptr() if t is not None else None)
    arr[i].mask_bits = l['mask_bits'].data_ptr()
    ns[i] = n
  prm = PruneRegrowParams(float(drop_fraction), int(grow_init_mode),
                          float(grow_init_div), int(momentum_reset_mode),
                          float(initial_acc_scale), int(bool(reinit_when_same)))
  need = lib.rigl_prune_regrow_workspace_bytes(ns, nl)
  ws = workspace(need, dev)
  counts = torch.zeros((nl, _lib.COUNTS_PER_LAYER), dtype=torch.int32,
                       device=dev)
  check(lib.rigl_prune_regrow(arr, nl, C.byref(prm), _ptr(counts), _ptr(ws),
                              ws.numel(), _stream()))
  return counts


def prune_regrow_selections(layer, drop_fraction, grow_init_mode=_lib.GROW_ZEROS, grow_init_div=1.0,
                            momentum_reset_mode=_lib.MOMRESET_GRAD, initial_acc_scale=0.0, reinit_when_same=False,
                            want_indices=True):
  """``prune_regrow`` on ONE layer (same dict), with the two selections read back (rigl_prune_regrow_selections): returns
  dict(counts int32[8], mask1_bits, mask2_bits (int32 bitmaps) and -- with ``want_indices`` -- idx1, idx2 int32[n]: the
  tensor's indices in tf.nn.top_k order of the drop / lifted grow score; the first counts[2] (n_keep) resp. counts[1]
  (n_prune) entries are the selected ones)."""
  lib = _lib.load()
  w = layer.get('w')
  ref = w if w is not None else layer['score_drop']
  n, dev = ref.numel(), ref.device
  arr = PruneRegrowLayer()
  arr.n = n
  arr.w = w.data_ptr() if w is not None else None
  for key in ('momentum', 'dense_grad', 'drop_noise', 'score_drop', 'score_grow', 'grow_values'):
    t = layer.get(key)
    _req(t, torch.float32, key, allow_none=True)
    setattr(arr, key, t.data_ptr() if t is not None else None)
  _req(layer['mask_bits'], torch.int32, 'mask_bits')
  arr.mask_bits = layer['mask_bits'].data_ptr()
  prm = PruneRegrowParams(float(drop_fraction), int(grow_init_mode), float(grow_init_div), int(momentum_reset_mode),
                          float(initial_acc_scale), int(bool(reinit_when_same)))
  words = n_mask_words(n)
  out = dict(counts=torch.zeros(_lib.COUNTS_PER_LAYER, dtype=torch.int32, device=dev),
             mask1_bits=torch.zeros(words, dtype=torch.int32, device=dev),
             mask2_bits=torch.zeros(words, dtype=torch.int32, device=dev))
  if want_indices:
    out['idx1'] = torch.empty(n, dtype=torch.int32, device=dev)
    out['idx2'] = torch.empty(n, dtype=torch.int32, device=dev)
  ws = workspace(lib.rigl_prune_regrow_selections_workspace_bytes(n), dev, 'k2sel')
  check(lib.rigl_prune_regrow_selections(C.byref(arr), C.byref(prm), _ptr(out['mask1_bits']), _ptr(out['mask2_bits']),
                                         _ptr(out.get('idx1')), _ptr(out.get('idx2')), _ptr(out['counts']), _ptr(ws),
                                         ws.numel(), _stream()))
  return out


def topk_mask(score, n_keep, out=None):
  """Bitmap of the n_keep largest scores (ties: lower flat index first)."""
  _req(score, torch.float32, 'score')
  n = score.numel()
  if out is None:
    out = torch.zeros(n_mask_words(n), dtype=torch.int32, device=score.device)
  lib = _lib.load()
  ns = (C.c_int64 * 1)(n)
  ws = workspace(lib.rigl_prune_regrow_workspace_bytes(ns, 1), score.device)
  check(lib.rigl_topk_mask(_ptr(score), n, int(n_keep), _ptr(out), _ptr(ws),
                           ws.numel(), _stream()))
  return out


def topk_mask_batched(items):
  """items: list of (score fp32 tensor, n_keep, mask_bits int32 tensor); all
  masks are rewritten in the same launches."""
  nl = len(items)
  if nl == 0:
    return
  lib = _lib.load()
  arr = (_lib.TopkLayer * nl)()
  ns = (C.c_int64 * nl)()
  for i, (score, n_keep, bits) in enumerate(items):
    _req(score, torch.float32, 'score')
    _req(bits, torch.int32, 'mask_bits')
    arr[i].score = score.data_ptr()
    arr[i].n = score.numel()
    arr[i].n_keep = int(n_keep)
    arr[i].mask_bits = bits.data_ptr()
    ns[i] = score.numel()
  ws = workspace(lib.rigl_prune_regrow_workspace_bytes(ns, nl), items[0][0].device)
  check(lib.rigl_topk_mask_batched(arr, nl, _ptr(ws), ws.numel(), _stream()))


# ----------------------------------------------------------------------------
# K3
# ----------------------------------------------------------------------------
def masked_sgd_momentum(w, grad, lr, momentum=None, mask_bits=None, mu=0.0,
                        weight_decay=0.0, grad_scale=1.0, nesterov=False,
                        w_shadow=None):
  _req(w, torch.float32, 'w')
  _req(grad, torch.float32, 'grad')
  _req(momentum, torch.float32, 'momentum', allow_none=True)
  _req(mask_bits, torch.int32, 'mask_bits', allow_none=True)
  _req(w_shadow, torch.bfloat16, 'w_shadow', allow_none=True)
  n = w.numel()
  if grad.numel() != n or (momentum is not None and momentum.numel() != n):
    raise ValueError('size mismatch')
  check(_lib.load().rigl_masked_sgd_momentum(
      n, _ptr(w), _ptr(momentum), _ptr(grad), _ptr(mask_bits), float(lr),
      float(mu), float(weight_decay), float(grad_scale), int(bool(nesterov)),
      _ptr(w_shadow), _stream()))


def pack_weights_batched(layers):
  """layers: list of (w fp32 [k*cout], mask_bits|None, k, cout, hwio|None,
  ohwi|None)."""
  nl = len(layers)
  if nl == 0:
    return
  arr = (PackLayer * nl)()
  for i, (w, bits, k, cout, hwio, ohwi) in enumerate(layers):
    _req(w, torch.float32, 'w')
    _req(bits, torch.int32, 'mask_bits', allow_none=True)
    _req(hwio, torch.bfloat16, 'hwio', allow_none=True)
    _req(ohwi, torch.bfloat16, 'ohwi', allow_none=True)
    if w.numel() != k * cout:
      raise ValueError('pack_weights: w has %d elements, k*cout=%d' %
                       (w.numel(), k * cout))
    arr[i].w = w.data_ptr()
    arr[i].mask_bits = bits.data_ptr() if bits is not None else None
    arr[i].hwio = hwio.data_ptr() if hwio is not None else None
    arr[i].ohwi = ohwi.data_ptr() if ohwi is not None else None
    arr[i].k = k
    arr[i].cout = cout
  check(_lib.load().rigl_pack_weights_batched(arr, nl, _stream()))


def pack_weights(w, mask_bits, k, cout, hwio=None, ohwi=None):
  pack_weights_batched([(w, mask_bits, k, cout, hwio, ohwi)])


# ----------------------------------------------------------------------------
# K1
# ----------------------------------------------------------------------------
# Dense-equivalent work of the K1 / K1d calls, counted on the host while ``work_count(True)`` is set: the
# algorithmic FLOPs / bytes that bench.py's roofline divides by the kernels' measured time.
WORK = {'fwd_macs': 0, 'dgrad_macs': 0, 'wgrad_macs': 0, 'depthwise_bytes': 0}
_work_on = False


def work_count(on=True):
  global _work_on
  _work_on = bool(on)
  if on:
    for k in WORK:
      WORK[k] = 0


def _count_macs(kind, d):
  if _work_on:
    WORK[kind] += d.n * d.ho * d.wo * d.kh * d.kw * d.cin * d.cout


def _count_depthwise(d):
  if _work_on:      # one bf16 tensor read + one written (fwd, dgrad) or two read (wgrad); weights are negligible
    WORK['depthwise_bytes'] += 2 * d.n * (d.h * d.w + d.ho * d.wo) * d.cin


# Plan-dependent sizes (workspace bytes, statistics parts) are asked of the library once per descriptor AND per knob
# generation: the C-side answers follow the run-time knobs (a kernel switched on after a layer's first call changes how
# many partial rows its forward writes; a stale, larger count would leave batch norm summing uninitialised rows).
# The generation is the LIBRARY's (rigl_tune_generation): a knob flipped through any other binding of the same process
# invalidates the cache too.


_GEN_WORD = None      # the library's counter, read in place (a ctypes call per lookup cost the launch-bound wrn22 step 0.5 ms)


def _plan_cached(d, name, fn):
  global _GEN_WORD
  if _GEN_WORD is None:
    _GEN_WORD = C.c_uint64.from_address(_lib.load().rigl_tune_generation_addr())
  gen = _GEN_WORD.value
  c = getattr(d, '_plan', None)
  if c is None or c[0] != gen:
    c = (gen, {})
    d._plan = c
  v = c[1].get(name)
  if v is None:
    v = c[1][name] = fn()
  return v


def conv_desc(n, h, w, cin, cout, kh, kw, stride, pad_top, pad_left, ho, wo):
  sh, sw = (stride, stride) if isinstance(stride, int) else stride
  return ConvDesc(n, h, w, cin, ho, wo, cout, kh, kw, sh, sw, pad_top,
                  pad_left)


def mfma_supported(d):
  """Shapes the implicit-GEMM MFMA kernels take (else the direct kernels)."""
  return d.cout % 8 == 0


def mfma_dgrad_supported(d):
  return d.cout % 8 == 0 and d.cin % 8 == 0


def conv_fwd(d, x, w_ohwi, y=None, force_ref=False, stats=False):
  """y[N,Ho,Wo,Cout] (bf16, NHWC memory) = conv(x, w).  With ``stats`` returns
  (y, partials): fp32 [parts, 2, Cout] batch-norm partial sums of y left by the
  conv epilogue (None where the MFMA path does not apply)."""
  _req(x, torch.bfloat16, 'x')
  _req(w_ohwi, torch.bfloat16, 'w_ohwi')
  _count_macs('fwd_macs', d)
  if y is None:
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.bfloat16,
                    device=x.device)
  _req(y, torch.bfloat16, 'y')
  lib = _lib.load()
  if force_ref or not mfma_supported(d):
    check(lib.rigl_conv2d_fwd_ref(C.byref(d), _ptr(x), _ptr(w_ohwi), _ptr(y),
                                  _stream()))
    return (y, None) if stats else y
  need = _plan_cached(d, 'ws_fwd', lambda: lib.rigl_conv2d_workspace_bytes(C.byref(d), 0))
  d._stats_parts = _plan_cached(d, 'stats_parts', lambda: lib.rigl_conv2d_stats_parts(C.byref(d)))
  ws = workspace(need, x.device) if need else None
  part = None
  if stats:
    parts = d._stats_parts
    part = torch.empty((parts, 2, d.cout), dtype=torch.float32, device=x.device)
  check(lib.rigl_masked_conv2d_fwd_stats(
      C.byref(d), _ptr(x), _ptr(w_ohwi), _ptr(y), _ptr(part),
      part.numel() if part is not None else 0, _ptr(ws),
      ws.numel() if ws is not None else 0, _stream()))
  return (y, part) if stats else y


def conv_fwd_takes_bn_input(d):
  """Does this layer's forward take the batch norm in front of it on its operand load (rigl_conv2d_fwd_takes_bn_input)?"""
  if not mfma_supported(d):
    return False
  return bool(_plan_cached(d, 'fwd_bn_input', lambda: int(_lib.load().rigl_conv2d_fwd_takes_bn_input(C.byref(d)))))


def conv_fwd_bnrelu(d, x_pre, saved, w_ohwi, a_out, y=None, stats=False):
  """y = conv(relu(bn(x_pre)), w) with the batch norm's apply pass on the operand load (rigl_masked_conv2d_fwd_bnrelu):
  ``saved`` = fp32 [4, Cin] (mean, invstd, scale, shift: bn_statistics), ``a_out`` (bf16, the shape of x_pre) receives
  relu(bn(x_pre)) as a side output.  Returns y or (y, partials) like conv_fwd."""
  _req(x_pre, torch.bfloat16, 'x_pre')
  _req(a_out, torch.bfloat16, 'a_out')
  _req(w_ohwi, torch.bfloat16, 'w_ohwi')
  _req(saved, torch.float32, 'saved')
  if saved.numel() != 4 * d.cin or a_out.numel() != x_pre.numel() or x_pre.numel() != d.n * d.h * d.w * d.cin:
    raise ValueError('saved must be [4, Cin]; x_pre and a_out [n, h, w, Cin]')
  _count_macs('fwd_macs', d)
  if y is None:
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.bfloat16, device=x_pre.device)
  _req(y, torch.bfloat16, 'y')
  lib = _lib.load()
  part = None
  if stats:
    parts = _plan_cached(d, 'stats_parts', lambda: lib.rigl_conv2d_stats_parts(C.byref(d)))
    part = torch.empty((parts, 2, d.cout), dtype=torch.float32, device=x_pre.device)
  check(lib.rigl_masked_conv2d_fwd_bnrelu(C.byref(d), _ptr(x_pre), _ptr(saved[2]), _ptr(a_out), _ptr(w_ohwi), _ptr(y), _ptr(part),
                                          part.numel() if part is not None else 0, None, 0, _stream()))
  return (y, part) if stats else y


def conv_dgrad(d, dy, w_hwio, dx=None, force_ref=False, addend=None):
  """dx = conv2d_backprop_input(dy, w) (+ addend, fused into the epilogue: the
  gradient accumulation of a tensor that feeds this conv and a shortcut)."""
  _count_macs('dgrad_macs', d)
  _req(dy, torch.bfloat16, 'dy')
  _req(w_hwio, torch.bfloat16, 'w_hwio')
  if dx is None:
    dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.bfloat16,
                     device=dy.device)
  _req(dx, torch.bfloat16, 'dx')
  if addend is not None:
    _req(addend, torch.bfloat16, 'addend')
    if addend.numel() != dx.numel():
      raise ValueError('addend must have the shape of dx')
  lib = _lib.load()
  if force_ref or not mfma_dgrad_supported(d):
    check(lib.rigl_conv2d_dgrad_ref(C.byref(d), _ptr(dy), _ptr(w_hwio),
                                    _ptr(dx), _stream()))
    if addend is not None:
      dx += addend.view_as(dx)
    return dx
  check(lib.rigl_masked_conv2d_dgrad_acc(C.byref(d), _ptr(dy), _ptr(w_hwio),
                                         _ptr(addend), _ptr(dx), None, 0,
                                         _stream()))
  return dx


def conv_wgrad(d, x, dy, dw=None, force_ref=False):
  """Dense fp32 dW in HWIO order (flat [kh*kw*cin*cout])."""
  _count_macs('wgrad_macs', d)
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  if dw is None:
    dw = torch.empty(d.kh * d.kw * d.cin * d.cout, dtype=torch.float32,
                     device=x.device)
  _req(dw, torch.float32, 'dw')
  lib = _lib.load()
  if force_ref or not mfma_supported(d):
    check(lib.rigl_conv2d_wgrad_ref(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw),
                                    _stream()))
    return dw
  need = lib.rigl_conv2d_workspace_bytes(C.byref(d), 2)
  ws = workspace(need, x.device) if need else None
  check(lib.rigl_masked_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw),
                                     _ptr(ws),
                                     ws.numel() if ws is not None else 0,
                                     _stream()))
  return dw


def dgrad_stats_parts(d):
  """Row tiles of this layer's dgrad kernel = rows of the batch-norm partials its epilogue can leave
  (0: the layer's dgrad has no such epilogue)."""
  v = _plan_cached(d, 'dgrad_parts', lambda: int(_lib.load().rigl_conv2d_dgrad_stats_parts(C.byref(d))))
  return v


# A shortcut gradient handed over UNMASKED: data_ptr of the gradient tensor -> the 1-bit ReLU mask it still has to pass
# (left by workloads.nn._FusedBNFn.backward, taken by pruning_layers._MaskedConvForkFn.backward one autograd node later).
LAZY_ADDEND_BITS = {}


def conv_bwd_takes_masked_addend(d):
  """Does this layer's one-call backward take its addend unmasked + a 1-bit mask (rigl_masked_conv2d_bwd_masked)?"""
  return bool(_plan_cached(d, 'masked_addend', lambda: int(_lib.load().rigl_conv2d_bwd_takes_masked_addend(C.byref(d)))))


def conv_bwd(d, x, dy, w_hwio, dw, need_dx=True, addend=None, on_dw_ready=None, bn_fuse=None, addend_sub=None,
             addend_bits=None):
  """dW (into ``dw``, dense fp32) and -- when ``need_dx`` -- dX (+ ``addend``) of one conv with a single host
  transition and, for ordinary layers, a single launch (rigl_masked_conv2d_bwd) followed by the split-K reduce that
  completes dW.  ``on_dw_ready`` is called once dW's last kernel has been enqueued (the data-parallel exchange launches
  its buckets from there).  Returns dX or None.
  ``bn_fuse`` = dict(x=, saved=, relu=, relu_bits=) of the batch norm whose output this conv read: its backward
  reductions are computed in the dgrad epilogue (rigl_masked_conv2d_bwd_bn) and returned as
  ``bn_fuse['partials']`` (fp32 [parts, 2, Cin]) for bn_bwd; left unset where the layer's kernels cannot.
  ``addend_sub`` = (sh, sw): ``addend`` is the gradient of the subsampled view x[:, ::sh, ::sw, :] -- bf16
  [n, ceil(h / sh), ceil(w / sw), cin] -- added at those pixels only (rigl_masked_conv2d_bwd_sub)."""
  if addend_sub is not None and tuple(addend_sub) == (1, 1):
    addend_sub = None
  if addend_bits is not None:
    # ``addend_bits`` (uint8, one bit per element of ``addend``): the addend counts only where its bit is set
    if addend is None or not need_dx or bn_fuse is not None or addend_sub is not None or not conv_bwd_takes_masked_addend(d):
      raise ValueError('addend_bits needs an addend, dX and a layer whose backward takes a masked addend')
    _req(x, torch.bfloat16, 'x'); _req(dy, torch.bfloat16, 'dy'); _req(dw, torch.float32, 'dw')
    _req(addend, torch.bfloat16, 'addend'); _req(addend_bits, torch.uint8, 'addend_bits'); _req(w_hwio, torch.bfloat16, 'w_hwio')
    if addend.numel() != d.n * d.h * d.w * d.cin or addend_bits.numel() * 8 != addend.numel():
      raise ValueError('addend must have the shape of dx and addend_bits one bit per element of it')
    lib = _lib.load()
    _count_macs('wgrad_macs', d)
    _count_macs('dgrad_macs', d)
    need = _plan_cached(d, 'ws_wgrad', lambda: lib.rigl_conv2d_workspace_bytes(C.byref(d), 2))
    ws = workspace(need, x.device, 'wg') if need else None
    dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.bfloat16, device=dy.device)
    check(lib.rigl_masked_conv2d_bwd_masked(C.byref(d), _ptr(x), _ptr(dy), _ptr(w_hwio), _ptr(addend), _ptr(addend_bits), _ptr(dw),
                                            _ptr(dx), _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
    if on_dw_ready is not None:
      on_dw_ready()
    return dx
  if addend_sub is not None:
    if addend is None or not need_dx or bn_fuse is not None:
      raise ValueError('addend_sub needs an addend and dX, and does not combine with bn_fuse')
    if not (mfma_supported(d) and mfma_dgrad_supported(d)):
      raise ValueError('addend_sub needs cin % 8 == cout % 8 == 0')
  if bn_fuse is not None:
    bn_fuse.pop('partials', None)
    if not (need_dx and mfma_supported(d) and mfma_dgrad_supported(d) and dgrad_stats_parts(d) > 0):
      bn_fuse = None
  if not (mfma_supported(d) and (not need_dx or mfma_dgrad_supported(d))):
    conv_wgrad(d, x, dy, dw)
    if on_dw_ready is not None:
      on_dw_ready()
    return conv_dgrad(d, dy, w_hwio, addend=addend) if need_dx else None
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  _req(dw, torch.float32, 'dw')
  _req(addend, torch.bfloat16, 'addend', allow_none=True)
  lib = _lib.load()
  _count_macs('wgrad_macs', d)
  if need_dx:
    _count_macs('dgrad_macs', d)
  need = _plan_cached(d, 'ws_wgrad', lambda: lib.rigl_conv2d_workspace_bytes(C.byref(d), 2))
  ws = workspace(need, x.device, 'wg') if need else None
  dx = None
  if need_dx:
    _req(w_hwio, torch.bfloat16, 'w_hwio')
    dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.bfloat16, device=dy.device)
    if addend is not None and addend_sub is None and addend.numel() != dx.numel():
      raise ValueError('addend must have the shape of dx')
  if addend_sub is not None:
    sh, sw = addend_sub
    if addend.numel() != d.n * (-(-d.h // sh)) * (-(-d.w // sw)) * d.cin:
      raise ValueError('addend must be [n, ceil(h / sh), ceil(w / sw), cin]')
    check(lib.rigl_masked_conv2d_bwd_sub(
        C.byref(d), _ptr(x), _ptr(dy), _ptr(w_hwio), _ptr(addend), int(sh), int(sw), _ptr(dw), _ptr(dx), _ptr(ws),
        ws.numel() if ws is not None else 0, _stream()))
    if on_dw_ready is not None:
      on_dw_ready()
    return dx
  bn = None
  if bn_fuse is not None:
    bx, saved = bn_fuse['x'], bn_fuse['saved']
    _req(bx, torch.bfloat16, 'bn x')
    _req(saved, torch.float32, 'bn saved')
    bits = bn_fuse.get('relu_bits')
    _req(bits, torch.uint8, 'bn relu_bits', allow_none=True)
    if bx.numel() != dx.numel() or saved.numel() != 4 * d.cin:
      raise ValueError('bn_fuse: x must have the shape of dx and saved must be [4, Cin]')
    part = torch.empty((dgrad_stats_parts(d), 2, d.cin), dtype=torch.float32, device=dy.device)
    bn = _lib.BnReduceFuse(bx.data_ptr(), bits.data_ptr() if bits is not None else None, saved.data_ptr(),
                           int(bool(bn_fuse['relu'])), part.data_ptr(), part.numel())
    bn_fuse['partials'] = part
  check(lib.rigl_masked_conv2d_bwd_bn(
      C.byref(d), _ptr(x), _ptr(dy), _ptr(w_hwio), _ptr(addend), _ptr(dw), _ptr(dx), _ptr(ws),
      ws.numel() if ws is not None else 0, C.byref(bn) if bn is not None else None, _stream()))
  if on_dw_ready is not None:
    on_dw_ready()
  return dx


def conv_bwd_grid(d, x, dy, w_hwio, dw, on_dw_ready=None):
  """Backward of a strided 1x1 conv without padding with dX on the conv's own grid: returns bf16 [n, ho, wo, cin] -- the
  gradient at the pixels the conv read (zero elsewhere, never materialised) -- and writes the dense dW
  (rigl_masked_conv2d_bwd_grid).  The consumer is conv_bwd(..., addend=that, addend_sub=strides) of the tensor's other
  reader."""
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  _req(w_hwio, torch.bfloat16, 'w_hwio')
  _req(dw, torch.float32, 'dw')
  lib = _lib.load()
  _count_macs('wgrad_macs', d)
  _count_macs('dgrad_macs', ConvDesc(d.n, d.ho, d.wo, d.cin, d.ho, d.wo, d.cout, 1, 1, 1, 1, 0, 0))
  need = _plan_cached(d, 'ws_wgrad', lambda: lib.rigl_conv2d_workspace_bytes(C.byref(d), 2))
  ws = workspace(need, x.device, 'wg') if need else None
  dx = torch.empty((d.n, d.ho, d.wo, d.cin), dtype=torch.bfloat16, device=dy.device)
  check(lib.rigl_masked_conv2d_bwd_grid(C.byref(d), _ptr(x), _ptr(dy), _ptr(w_hwio), _ptr(dw), _ptr(dx), _ptr(ws),
                                        ws.numel() if ws is not None else 0, _stream()))
  if on_dw_ready is not None:
    on_dw_ready()
  return dx


# ----------------------------------------------------------------------------
# K1 in fp32 arithmetic (validation twin; the reference's --precision=float32)
# ----------------------------------------------------------------------------
def conv_fwd_f32(d, x, w_hwio, mask_bits=None, y=None):
  """y[N,Ho,Wo,Cout] fp32 = conv(x, mask * W): fp32 NHWC activations, the fp32 master weights (flat HWIO) and the
  mask bitmap (None: dense), on the fp32 MFMA (rigl_masked_conv2d_fwd_f32)."""
  _req(x, torch.float32, 'x')
  _req(w_hwio, torch.float32, 'w_hwio')
  _req(mask_bits, torch.int32, 'mask_bits', allow_none=True)
  _count_macs('fwd_macs', d)
  if y is None:
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
  _req(y, torch.float32, 'y')
  check(_lib.load().rigl_masked_conv2d_fwd_f32(C.byref(d), _ptr(x), _ptr(w_hwio), _ptr(mask_bits), _ptr(y), _stream()))
  return y


def conv_bwd_f32(d, x, dy, w_hwio, mask_bits, dw, need_dx=True, addend=None, on_dw_ready=None):
  """fp32 twin of conv_bwd: dense dW (fp32 HWIO, overwritten) and -- when ``need_dx`` -- dX (+ ``addend``)."""
  _req(x, torch.float32, 'x')
  _req(dy, torch.float32, 'dy')
  _req(dw, torch.float32, 'dw')
  _req(addend, torch.float32, 'addend', allow_none=True)
  _req(mask_bits, torch.int32, 'mask_bits', allow_none=True)
  lib = _lib.load()
  _count_macs('wgrad_macs', d)
  need = _plan_cached(d, 'ws_wgrad_f32', lambda: lib.rigl_conv2d_wgrad_f32_workspace_bytes(C.byref(d)))
  ws = workspace(need, x.device, 'wg32') if need else None
  check(lib.rigl_masked_conv2d_wgrad_f32(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws),
                                         ws.numel() if ws is not None else 0, _stream()))
  if on_dw_ready is not None:
    on_dw_ready()
  if not need_dx:
    return None
  _count_macs('dgrad_macs', d)
  _req(w_hwio, torch.float32, 'w_hwio')
  dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.float32, device=dy.device)
  if addend is not None and addend.numel() != dx.numel():
    raise ValueError('addend must have the shape of dx')
  check(lib.rigl_masked_conv2d_dgrad_f32(C.byref(d), _ptr(dy), _ptr(w_hwio), _ptr(mask_bits), _ptr(addend), _ptr(dx),
                                         _stream()))
  return dx


# ----------------------------------------------------------------------------
# K1d depthwise
# ----------------------------------------------------------------------------
def depthwise_fwd(d, x, w, stats=False):
  """y = depthwise conv(x, w).  With ``stats`` returns (y, partials): fp32 [parts, 2, C] batch-norm partial sums of y left
  by the kernel's epilogue (None for shapes without one)."""
  _count_depthwise(d)
  _req(x, torch.bfloat16, 'x')
  _req(w, torch.float32, 'w')
  y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.bfloat16, device=x.device)
  lib = _lib.load()
  if stats:
    parts = _plan_cached(d, 'dw_parts', lambda: int(lib.rigl_depthwise_conv2d_stats_parts(C.byref(d))))
    if parts > 0:
      part = torch.empty((parts, 2, d.cout), dtype=torch.float32, device=x.device)
      check(lib.rigl_depthwise_conv2d_fwd_stats(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(part), part.numel(), _stream()))
      return y, part
  check(lib.rigl_depthwise_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _stream()))
  return (y, None) if stats else y


def depthwise_dgrad(d, dy, w):
  _count_depthwise(d)
  _req(dy, torch.bfloat16, 'dy')
  _req(w, torch.float32, 'w')
  dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.bfloat16, device=dy.device)
  check(_lib.load().rigl_depthwise_conv2d_dgrad(C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _stream()))
  return dx


def depthwise_wgrad(d, x, dy, dw):
  _count_depthwise(d)
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  _req(dw, torch.float32, 'dw')
  lib = _lib.load()
  ws = workspace(lib.rigl_depthwise_conv2d_workspace_bytes(C.byref(d)), x.device)
  check(lib.rigl_depthwise_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel(), _stream()))
  return dw


# ----------------------------------------------------------------------------
# fused batch-norm (+ residual) (+ ReLU)
# ----------------------------------------------------------------------------
def bn_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu,
           residual=None, partials=None, want_relu_bits=False):
  """x [..., C] bf16 contiguous.  Returns (y, saved) with saved = fp32 [4, C]
  (mean, invstd, scale, shift).  ``partials`` (fp32 [parts, 2, C], from
  conv_fwd(stats=True)) replaces the statistics pass over x.  With
  ``want_relu_bits`` returns (y, saved, bits): uint8 [numel/8], bit j of byte i =
  (y[8i+j] > 0), the ReLU mask the backward needs instead of y."""
  _req(x, torch.bfloat16, 'x')
  _req(residual, torch.bfloat16, 'residual', allow_none=True)
  for t, nm in ((gamma, 'gamma'), (beta, 'beta')):
    _req(t, torch.float32, nm)
  c = x.shape[-1]
  m = x.numel() // c
  lib = _lib.load()
  y = torch.empty_like(x)
  saved = torch.empty((4, c), dtype=torch.float32, device=x.device)
  bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) \
      if (want_relu_bits and relu) else None
  if partials is not None:
    _req(partials, torch.float32, 'partials')
    if partials.dim() != 3 or partials.shape[1] != 2 or partials.shape[2] != c:
      raise ValueError('partials must be [parts, 2, C]')
    ws = None
  else:
    ws = workspace(lib.rigl_bn_workspace_bytes(m, c), x.device)
  check(lib.rigl_bn_fwd_stats(
      m, c, _ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta),
      _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
      int(bool(relu)), _ptr(y), _ptr(saved[0]), _ptr(saved[1]), _ptr(saved[2]),
      _ptr(saved[3]), _ptr(partials),
      partials.shape[0] if partials is not None else 0, _ptr(bits), _ptr(ws),
      ws.numel() if ws is not None else 0, _stream()))
  return (y, saved, bits) if want_relu_bits else (y, saved)


def bn_bwd(x, y, dy, gamma, saved, relu, dgamma, dbeta, want_dres=False,
           relu_bits=None, partials=None):
  """Returns (dx, dres|None); dgamma / dbeta (fp32 [C]) are overwritten.  The
  ReLU mask comes from ``relu_bits`` (bn_fwd(want_relu_bits=True)), else ``y``,
  else it is recomputed from x.  ``partials`` (fp32 [parts, 2, C]: sum dz, sum dz * xhat per row
  tile, left by the dgrad epilogue that produced ``dy`` -- conv_bwd(bn_fuse=...)) replaces the
  reduction pass over dy and x."""
  _req(relu_bits, torch.uint8, 'relu_bits', allow_none=True)
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  _req(y, torch.bfloat16, 'y', allow_none=True)
  c = x.shape[-1]
  m = x.numel() // c
  lib = _lib.load()
  dx = torch.empty_like(x)
  dres = torch.empty_like(x) if want_dres else None
  ws = workspace(lib.rigl_bn_workspace_bytes(m, c), x.device)
  _req(partials, torch.float32, 'partials', allow_none=True)
  if partials is not None and (partials.dim() != 3 or partials.shape[1] != 2 or partials.shape[2] != c):
    raise ValueError('partials must be [parts, 2, %d]' % c)
  check(lib.rigl_bn_bwd_stats(m, c, _ptr(x), _ptr(y), _ptr(relu_bits), _ptr(dy), _ptr(gamma),
                              _ptr(saved[0]), _ptr(saved[1]), _ptr(saved[2]),
                              _ptr(saved[3]), int(bool(relu)), _ptr(dx), _ptr(dres),
                              _ptr(dgamma), _ptr(dbeta), _ptr(partials),
                              partials.shape[0] if partials is not None else 0, _ptr(ws), ws.numel(),
                              _stream()))
  return dx, dres


# ----------------------------------------------------------------------------
# stateless random tensors (TensorFlow bit layout)
# ----------------------------------------------------------------------------
def stateless_random(n, seed0, seed1, dist, scale=1.0, shift=0.0, device=None,
                     out=None):
  """fp32 [n] = tf.random.stateless_{uniform|normal}([n], seed=[seed0, seed1])
  * scale + shift.  seed0 / seed1 are int32 (Python ints are wrapped)."""
  def i32(v):
    v = int(v) & 0xFFFFFFFF
    return v - (1 << 32) if v >= 1 << 31 else v
  if dist not in ('uniform', 'normal'):
    raise ValueError('dist must be "uniform" or "normal"')
  if out is None:
    out = torch.empty(int(n), dtype=torch.float32, device=device)
  _req(out, torch.float32, 'out')
  check(_lib.load().rigl_stateless_random(_ptr(out), out.numel(), i32(seed0), i32(seed1),
                                          1 if dist == 'normal' else 0,
                                          float(scale), float(shift), _stream()))
  return out


def stateless_random_batched(items, device):
  """items: [(n, seed0, seed1, 'uniform'|'normal', scale, shift)] -> list of fp32 [n] tensors (views of one
  allocation, each starting on a 256-byte boundary), filled by ONE launch (rigl_stateless_random_batched)."""
  def i32(v):
    v = int(v) & 0xFFFFFFFF
    return v - (1 << 32) if v >= 1 << 31 else v
  if not items:
    return []
  offs, total = [], 0
  for it in items:
    offs.append(total)
    total += (int(it[0]) + 63) // 64 * 64
  buf = torch.empty(max(total, 1), dtype=torch.float32, device=device)
  arr = (_lib.RandomItem * len(items))()
  outs = []
  for i, (n, s0, s1, dist, scale, shift) in enumerate(items):
    if dist not in ('uniform', 'normal'):
      raise ValueError('dist must be "uniform" or "normal"')
    t = buf[offs[i]:offs[i] + int(n)]
    outs.append(t)
    arr[i].out = t.data_ptr() if int(n) else None
    arr[i].n = int(n)
    arr[i].seed0, arr[i].seed1 = i32(s0), i32(s1)
    arr[i].dist = 1 if dist == 'normal' else 0
    arr[i].scale, arr[i].shift = float(scale), float(shift)
  check(_lib.load().rigl_stateless_random_batched(arr, len(items), _stream()))
  return outs


# ----------------------------------------------------------------------------
# max pooling (glue)
# ----------------------------------------------------------------------------
def maxpool_fwd(d, x):
  """Returns (y, argmax) for the pooling geometry ``d`` (conv_desc with
  cin == cout); argmax is uint8, one byte per output element."""
  _req(x, torch.bfloat16, 'x')
  y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.bfloat16, device=x.device)
  arg = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.uint8, device=x.device)
  check(_lib.load().rigl_maxpool_fwd(C.byref(d), _ptr(x), _ptr(y), _ptr(arg), _stream()))
  return y, arg


def maxpool_bwd(d, dy, arg):
  _req(dy, torch.bfloat16, 'dy')
  _req(arg, torch.uint8, 'argmax')
  dx = torch.empty((d.n, d.h, d.w, d.cin), dtype=torch.bfloat16, device=dy.device)
  check(_lib.load().rigl_maxpool_bwd(C.byref(d), _ptr(dy), _ptr(arg), _ptr(dx), _stream()))
  return dx


def bn_statistics(x, gamma, beta, running_mean, running_var, momentum, eps, partials=None):
  """The statistics half of bn_fwd (no apply pass): returns saved = fp32 [4, C] (mean, invstd, scale, shift) and updates
  the moving averages (rigl_bn_fwd_statistics)."""
  _req(x, torch.bfloat16, 'x')
  c = x.shape[-1]
  m = x.numel() // c
  lib = _lib.load()
  saved = torch.empty((4, c), dtype=torch.float32, device=x.device)
  if partials is not None:
    _req(partials, torch.float32, 'partials')
    ws = None
  else:
    ws = workspace(lib.rigl_bn_workspace_bytes(m, c), x.device)
  check(lib.rigl_bn_fwd_statistics(
      m, c, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
      _ptr(saved[0]), _ptr(saved[1]), _ptr(saved[2]), _ptr(saved[3]), _ptr(partials),
      partials.shape[0] if partials is not None else 0, _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
  return saved


def bn_add_bn_fwd(x, x2, saved2, gamma, beta, running_mean, running_var, momentum, eps, relu, partials=None):
  """relu?(bn(x) + bn2(x2)) in one apply pass; ``saved2`` = bn_statistics(x2, ...).  Returns (y, saved, relu_bits|None)."""
  _req(x, torch.bfloat16, 'x')
  _req(x2, torch.bfloat16, 'x2')
  if x2.shape != x.shape:
    raise ValueError('x2 must have the shape of x')
  c = x.shape[-1]
  m = x.numel() // c
  lib = _lib.load()
  y = torch.empty_like(x)
  saved = torch.empty((4, c), dtype=torch.float32, device=x.device)
  bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if relu else None
  if partials is not None:
    _req(partials, torch.float32, 'partials')
    ws = None
  else:
    ws = workspace(lib.rigl_bn_workspace_bytes(m, c), x.device)
  check(lib.rigl_bn_add_bn_fwd(
      m, c, _ptr(x), _ptr(x2), _ptr(saved2[2]), _ptr(saved2[3]), _ptr(gamma), _ptr(beta), _ptr(running_mean),
      _ptr(running_var), float(momentum), float(eps), int(bool(relu)), _ptr(y), _ptr(saved[0]), _ptr(saved[1]),
      _ptr(saved[2]), _ptr(saved[3]), _ptr(partials), partials.shape[0] if partials is not None else 0, _ptr(bits),
      _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
  return y, saved, bits


def bn_add_bn_bwd(x, x2, relu_bits, dy, gamma, saved, gamma2, saved2, dgamma, dbeta, dgamma2, dbeta2):
  """Gradients of relu(bn(x) + bn2(x2)) w.r.t. x and x2 (returns (dx, dx2)); the four parameter gradients are overwritten."""
  _req(x, torch.bfloat16, 'x')
  _req(x2, torch.bfloat16, 'x2')
  _req(dy, torch.bfloat16, 'dy')
  _req(relu_bits, torch.uint8, 'relu_bits')
  c = x.shape[-1]
  m = x.numel() // c
  lib = _lib.load()
  ws = workspace(lib.rigl_bn_add_bn_bwd_workspace_bytes(m, c), x.device)
  dx, dx2 = torch.empty_like(x), torch.empty_like(x2)
  check(lib.rigl_bn_add_bn_bwd(m, c, _ptr(x), _ptr(x2), _ptr(relu_bits), _ptr(dy), _ptr(gamma), _ptr(saved[0]), _ptr(saved[1]),
                               _ptr(gamma2), _ptr(saved2[0]), _ptr(saved2[1]), _ptr(dx), _ptr(dx2), _ptr(dgamma), _ptr(dbeta),
                               _ptr(dgamma2), _ptr(dbeta2), _ptr(ws), ws.numel(), _stream()))
  return dx, dx2


def bn_relu_maxpool_fwd(d, x, gamma, beta, running_mean, running_var, momentum, eps, partials=None):
  """maxpool(relu(batch_norm(x))) for the pooling geometry ``d`` without the activated tensor
  (rigl_bn_fwd_statistics + rigl_bn_relu_maxpool_fwd).  Returns (y, argmax, saved) -- saved = fp32 [4, C]
  (mean, invstd, scale, shift) as bn_fwd returns it."""
  saved = bn_statistics(x, gamma, beta, running_mean, running_var, momentum, eps, partials=partials)
  lib = _lib.load()
  y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.bfloat16, device=x.device)
  arg = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.uint8, device=x.device)
  check(lib.rigl_bn_relu_maxpool_fwd(C.byref(d), _ptr(x), _ptr(saved[2]), _ptr(saved[3]), _ptr(y), _ptr(arg), _stream()))
  return y, arg, saved


def bn_relu_maxpool_bwd(d, x, dy, arg, gamma, saved, dgamma, dbeta):
  """Gradient of bn_relu_maxpool_fwd w.r.t. x; dgamma / dbeta (fp32 [C]) are overwritten."""
  _req(x, torch.bfloat16, 'x')
  _req(dy, torch.bfloat16, 'dy')
  _req(arg, torch.uint8, 'argmax')
  lib = _lib.load()
  ws = workspace(lib.rigl_bn_relu_maxpool_bwd_workspace_bytes(C.byref(d)), x.device)
  dx = torch.empty_like(x)
  check(lib.rigl_bn_relu_maxpool_bwd(C.byref(d), _ptr(x), _ptr(dy), _ptr(arg), _ptr(gamma), _ptr(saved[0]), _ptr(saved[1]),
                                     _ptr(saved[2]), _ptr(saved[3]), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                     ws.numel(), _stream()))
  return dx


# ----------------------------------------------------------------------------
# classifier head glue
# ----------------------------------------------------------------------------
def global_avgpool_fwd(x):
  """[N, H, W, C] bf16 -> [N, C] bf16, fp32 accumulation."""
  _req(x, torch.bfloat16, 'x')
  n, h, w, c = x.shape
  y = torch.empty((n, c), dtype=torch.bfloat16, device=x.device)
  check(_lib.load().rigl_global_avgpool_fwd(n, h * w, c, _ptr(x), _ptr(y), _stream()))
  return y


def global_avgpool_bwd(dy, h, w):
  _req(dy, torch.bfloat16, 'dy')
  n, c = dy.shape
  dx = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dy.device)
  check(_lib.load().rigl_global_avgpool_bwd(n, h * w, c, _ptr(dy), _ptr(dx), _stream()))
  return dx


def softmax_xent(logits, labels, label_smoothing=0.0, grad_scale=None, want_grad=True):
  """Per-row cross entropy (fp32 [rows]) of bf16 logits [rows, classes] against int64 labels with label
  smoothing, and the gradient of ``grad_scale * sum(rows)`` w.r.t. the logits (bf16; default scale 1/rows)."""
  _req(logits, torch.bfloat16, 'logits')
  _req(labels, torch.int64, 'labels')
  rows, k = logits.shape
  if labels.numel() != rows:
    raise ValueError('softmax_xent: %d labels for %d rows' % (labels.numel(), rows))
  loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
  dz = torch.empty_like(logits) if want_grad else None
  check(_lib.load().rigl_softmax_xent(rows, k, _ptr(logits), _ptr(labels), float(label_smoothing),
                                      1.0 / rows if grad_scale is None else float(grad_scale), _ptr(loss), _ptr(dz),
                                      _stream()))
  return loss, dz


# ----------------------------------------------------------------------------
# profiling
# ----------------------------------------------------------------------------
def tune_set(key, value):
  """Process-wide kernel-selection knob (rigl_tune_set).  The plan-dependent sizes cached on descriptors (_plan_cached)
  are keyed on the knob generation, so a descriptor made before the call asks the library again."""
  check(_lib.load().rigl_tune_set(key.encode(), int(value)))


def tune_unset(key):
  """Back to the knob's RIGL_<KEY> environment variable / built-in default (rigl_tune_unset)."""
  check(_lib.load().rigl_tune_unset(key.encode()))


def tune_get(key, default=-1):
  return int(_lib.load().rigl_tune_get(key.encode(), int(default)))


def mfma_peak_probe(device='cuda:0', blocks=2048, iters=2000, reps=5):
  """Dense bf16 MFMA rate of this box in TFLOP/s (register-only MFMA chains, best of ``reps``)."""
  lib = _lib.load()
  sink = torch.zeros(blocks * 256, dtype=torch.float32, device=device)
  best = 0.0
  for r in range(reps + 1):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(lib.rigl_probe_mfma_bf16(blocks, iters, _ptr(sink), _stream()))
    e.record()
    e.synchronize()
    if r:
      best = max(best, blocks * 4 * iters * 8 * 32768.0 / (s.elapsed_time(e) * 1e-3) / 1e12)
  return best


def prof_enable(on=True):
  check(_lib.load().rigl_prof_enable(int(bool(on))))


def prof_collect_launches(cap=1 << 16):
  """[(kind name, (h, w, cin, cout, k, stride), ms)] of every timed dispatch since the last collect, in launch order
  (rigl_prof_collect_launches; consumes the events like prof_collect)."""
  arr = (_lib.ProfLaunch * cap)()
  n = C.c_int64(0)
  check(_lib.load().rigl_prof_collect_launches(arr, cap, C.byref(n)))
  if n.value > cap:
    raise RuntimeError('prof_collect_launches: %d timed dispatches, room for %d (the rest were consumed and dropped): '
                       'collect more often or pass a larger cap' % (n.value, cap))
  return [(_lib.PROF_KINDS[arr[i].kind], tuple(arr[i].tag), float(arr[i].ms)) for i in range(min(n.value, cap))]


def prof_collect():
  """{kind: (milliseconds, launches)} accumulated since the last collect."""
  ms = (C.c_double * len(_lib.PROF_KINDS))()
  cnt = (C.c_int64 * len(_lib.PROF_KINDS))()
  check(_lib.load().rigl_prof_collect(ms, cnt))
  return {k: (ms[i], cnt[i]) for i, k in enumerate(_lib.PROF_KINDS)}
