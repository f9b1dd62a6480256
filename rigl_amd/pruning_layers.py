"""The masked-layer boundary: ``sparse_conv2d`` / ``sparse_fully_connected``.

Mirrors rigl/imagenet_resnet/pruning_layers.py:72-248 (same argument names,
same ``sparsity_technique`` values, same ValueErrors) on top of the HIP
kernels: ``sparsity_technique='threshold'`` creates `{scope}/weights` and
`{scope}/mask` exactly like contrib's ``masked_conv2d`` (mask initialised to
ones, y = conv(x, mask * W)); ``'baseline'`` creates a dense layer.

Activations are NHWC bf16 tensors ([N,H,W,C] contiguous) -- the reference's
``channels_last`` default and bfloat16 scope (imagenet_train_eval.py:92-98,
549-552).  Weights are fp32 HWIO variables of the graph (variables.py).  The
autograd bridge calls the C ABI: forward = rigl_masked_conv2d_fwd, backward =
rigl_masked_conv2d_dgrad + rigl_masked_conv2d_wgrad; the DENSE dW goes
straight into the layer's slice of the gradient arena (RigL needs it dense,
sparse_optimizers_base.py:478-485) -- there is no PyTorch fallback.
"""
import math
import os

import numpy as np
import torch

from rigl_amd import ops
from rigl_amd import variables as V


def _same_pad(size, k, stride):
  """TF 'SAME': out = ceil(size/stride), pad_total = max((out-1)*s + k - size, 0),
  pad_begin = pad_total // 2 (the extra pixel goes at the end)."""
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  return out, total // 2


def _valid_out(size, k, stride):
  return (size - k) // stride + 1


# RIGL_BN_FUSE_BWD=1: a batch norm's backward reductions ride in the dgrad epilogue of its output's only consumer
# (rigl_masked_conv2d_bwd_bn).  OFF by default: measured on ResNet-50 at batch 128 the reduction pass it removes costs
# 10-17 us per layer and the epilogue work it adds 15-19 us (profiles/r2/README.md) -- 13.29 vs 13.30 ms per step
# with K1's share of it up by 1 ms.
_BN_FUSE_BWD = os.environ.get('RIGL_BN_FUSE_BWD', '0') == '1'
# RIGL_CONV_PAIR=0: the first block of a group runs projection and conv1 as two autograd nodes (read once; tests override
# the module attribute).  The pair node carries no batch-norm holder: with RIGL_BN_FUSE_BWD=1 it steps aside for the fork path.
_CONV_PAIR = os.environ.get('RIGL_CONV_PAIR', '1') != '0'
# ... only for activations of at most this many MB (the large early tensors make the dgrad launch HBM-bound already)
_BN_FUSE_MAX_BYTES = float(os.environ.get('RIGL_BN_FUSE_MAX_MB', '1e9')) * 1e6


def _bn_source(x):
  """The batch norm that produced ``x`` (workloads.nn.BatchNorm leaves a holder on its output): its backward
  reductions can ride in this conv's dgrad epilogue.  Counted, because that is only right for the SOLE consumer."""
  h = getattr(x, 'bn_ctx', None) if _BN_FUSE_BWD else None
  if h is not None:
    h.attached += 1
  return h


def _bn_fuse_request(holder, need_dx):
  if holder is None or not need_dx or holder.attached != 1 or holder.x is None:
    return None
  if holder.x.numel() * 2 > _BN_FUSE_MAX_BYTES:
    return None
  return dict(x=holder.x, saved=holder.saved, relu=holder.relu, relu_bits=holder.bits)


def _bn_fuse_publish(holder, req, dx):
  if req is not None and req.get('partials') is not None and dx is not None:
    holder.partials, holder.dx_ptr = req['partials'], dx.data_ptr()


def _mask_bits(lv):
  """The layer's mask bitmap for the fp32 kernels (they read the master weights and apply the mask on the fly)."""
  return lv.mask.bits if lv.mask is not None else None


class _MaskedConvFn(torch.autograd.Function):
  """y = conv(x, mask*W) with dense dW written into lv.weights.grad."""

  @staticmethod
  def forward(ctx, x, lv, desc, need_dx, want_stats=False, bn_holder=None, pending=None):
    ctx.lv, ctx.desc, ctx.need_dx = lv, desc, need_dx
    ctx.bn_holder = bn_holder
    ctx.save_for_backward(x)
    if pending is not None:
      # x is the still EMPTY output of a batch norm that only finalised its statistics (workloads.nn.BatchNorm, consumer=):
      # this conv reads the pre-batch-norm tensor, applies scale / shift / ReLU on its operand load and fills x on the way
      out = ops.conv_fwd_bnrelu(desc, pending.x, pending.saved, lv.ohwi, x, stats=want_stats)
      if not want_stats:
        return out
      y, part = out
      ctx.mark_non_differentiable(part)
      ctx.set_materialize_grads(False)
      return y, part
    if x.dtype == torch.float32:         # --precision=float32: the fp32 validation kernels, no statistics epilogue
      y = ops.conv_fwd_f32(desc, x, lv.weights.data.view(-1), _mask_bits(lv))
      if not want_stats:
        return y
      part = torch.empty(0, device=x.device)
      ctx.mark_non_differentiable(part)
      ctx.set_materialize_grads(False)
      return y, part
    if not want_stats:
      return ops.conv_fwd(desc, x, lv.ohwi)
    y, part = ops.conv_fwd(desc, x, lv.ohwi, stats=True)
    if part is None:
      part = torch.empty(0, device=x.device)
    ctx.mark_non_differentiable(part)
    ctx.set_materialize_grads(False)     # no zero-filled "gradient" for the statistics output
    return y, part

  @staticmethod
  def backward(ctx, dy, _dpart=None):
    (x,) = ctx.saved_tensors
    lv, d = ctx.lv, ctx.desc
    if dy is None:                       # output unused: zero gradient
      dy = torch.zeros((d.n, d.ho, d.wo, d.cout), dtype=x.dtype, device=x.device)
    dy = dy.contiguous()
    # dense dL/d(mask*W), fp32 HWIO, into this layer's slice of the G arena, and dX -- one call
    sync = getattr(lv.weights.graph, 'grad_sync', None)
    ready = (lambda: sync.notify_layer_grad_ready(lv.weights)) if sync is not None else None   # DP: overlap the all-reduce
    if x.dtype == torch.float32:
      dx = ops.conv_bwd_f32(d, x, dy, lv.weights.data.view(-1), _mask_bits(lv), lv.weights.grad.view(-1),
                            need_dx=ctx.need_dx, on_dw_ready=ready)
      return dx, None, None, None, None, None, None
    req = _bn_fuse_request(ctx.bn_holder, ctx.need_dx)
    dx = ops.conv_bwd(d, x, dy, lv.hwio, lv.weights.grad.view(-1), need_dx=ctx.need_dx, on_dw_ready=ready, bn_fuse=req)
    _bn_fuse_publish(ctx.bn_holder, req, dx)
    return dx, None, None, None, None, None, None


class _MaskedConvForkFn(torch.autograd.Function):
  """(y, x') = (conv(x, mask*W), x): the conv plus an alias of its input for the
  tensor's OTHER consumer (a residual shortcut, or the next conv reading the
  same block input).  Backward folds the alias' gradient into the dgrad
  epilogue -- dx = conv2d_backprop_input(dy, mask*W) + dx' -- instead of the
  separate AddN pass autodiff would emit (rigl_masked_conv2d_dgrad_acc)."""

  @staticmethod
  def forward(ctx, x, lv, desc, want_stats=False, bn_holder=None):
    ctx.lv, ctx.desc = lv, desc
    ctx.bn_holder = bn_holder
    ctx.save_for_backward(x)
    if x.dtype == torch.float32:
      y = ops.conv_fwd_f32(desc, x, lv.weights.data.view(-1), _mask_bits(lv))
      if not want_stats:
        return y, x.view_as(x)
      part = torch.empty(0, device=x.device)
      ctx.mark_non_differentiable(part)
      ctx.set_materialize_grads(False)
      return y, x.view_as(x), part
    if not want_stats:
      return ops.conv_fwd(desc, x, lv.ohwi), x.view_as(x)
    y, part = ops.conv_fwd(desc, x, lv.ohwi, stats=True)
    if part is None:
      part = torch.empty(0, device=x.device)
    ctx.mark_non_differentiable(part)
    ctx.set_materialize_grads(False)
    return y, x.view_as(x), part

  @staticmethod
  def backward(ctx, dy, dalias, _dpart=None):
    (x,) = ctx.saved_tensors
    lv, d = ctx.lv, ctx.desc
    if dy is None:
      dy = torch.zeros((d.n, d.ho, d.wo, d.cout), dtype=x.dtype, device=x.device)
    dy = dy.contiguous()
    if dalias is not None:
      dalias = dalias.contiguous()
    sync = getattr(lv.weights.graph, 'grad_sync', None)
    ready = (lambda: sync.notify_layer_grad_ready(lv.weights)) if sync is not None else None
    if x.dtype == torch.float32:
      dx = ops.conv_bwd_f32(d, x, dy, lv.weights.data.view(-1), _mask_bits(lv), lv.weights.grad.view(-1),
                            need_dx=True, addend=dalias, on_dw_ready=ready)
      return dx, None, None, None, None
    # the alias' gradient may arrive unmasked with the ReLU bits it still has to pass (workloads.nn._FusedBNFn, lazy_res_grad)
    lazy = ops.LAZY_ADDEND_BITS.pop(dalias.data_ptr(), None) if dalias is not None else None
    if lazy is not None:
      dx = ops.conv_bwd(d, x, dy, lv.hwio, lv.weights.grad.view(-1), need_dx=True, addend=dalias, addend_bits=lazy[0],
                        on_dw_ready=ready)
      return dx, None, None, None, None
    # dx = this conv's dgrad + the alias' gradient is the COMPLETE gradient of the forked tensor: the producing batch
    # norm's reductions are taken on it
    req = _bn_fuse_request(ctx.bn_holder, True)
    dx = ops.conv_bwd(d, x, dy, lv.hwio, lv.weights.grad.view(-1), need_dx=True, addend=dalias, on_dw_ready=ready, bn_fuse=req)
    _bn_fuse_publish(ctx.bn_holder, req, dx)
    return dx, None, None, None, None


class _MaskedConvPairFn(torch.autograd.Function):
  """(y_s, y_m) = (conv_s(x), conv_m(x)): the two readers of a tensor where conv_s is a STRIDED 1x1 conv without padding
  -- the projection shortcut of a ResNet group's first block next to its conv1 (resnet_model.py:456-501).  conv_s only
  reads the pixels (i * sh, j * sw), so dL/dx through it is zero everywhere else.  As two autograd nodes that gradient is
  a full-size tensor (three quarters zeros at stride 2) written by a strided dgrad and read back by the other conv's
  accumulating epilogue; as ONE node it is computed on the [n, ho, wo] grid as the dgrad of a stride-1 1x1 conv and
  handed to conv_m's dgrad epilogue compact (rigl_masked_conv2d_bwd_sub): dx = dgrad_m(dy_m) + scatter(dgrad_s(dy_s)).
  Both dense dW go straight into their gradient-arena slices, as in _MaskedConvFn."""

  @staticmethod
  def forward(ctx, x, lv_s, d_s, lv_m, d_m, want_stats):
    ctx.lv_s, ctx.d_s, ctx.lv_m, ctx.d_m = lv_s, d_s, lv_m, d_m
    ctx.save_for_backward(x)
    if not want_stats:
      return ops.conv_fwd(d_s, x, lv_s.ohwi), ops.conv_fwd(d_m, x, lv_m.ohwi)
    ys, ps = ops.conv_fwd(d_s, x, lv_s.ohwi, stats=True)
    ym, pm = ops.conv_fwd(d_m, x, lv_m.ohwi, stats=True)
    ps = ps if ps is not None else torch.empty(0, device=x.device)
    pm = pm if pm is not None else torch.empty(0, device=x.device)
    ctx.mark_non_differentiable(ps, pm)
    ctx.set_materialize_grads(False)
    return ys, ym, ps, pm

  @staticmethod
  def backward(ctx, dys, dym, _dps=None, _dpm=None):
    (x,) = ctx.saved_tensors
    lv_s, d_s, lv_m, d_m = ctx.lv_s, ctx.d_s, ctx.lv_m, ctx.d_m
    if dys is None:
      dys = torch.zeros((d_s.n, d_s.ho, d_s.wo, d_s.cout), dtype=torch.bfloat16, device=x.device)
    if dym is None:
      dym = torch.zeros((d_m.n, d_m.ho, d_m.wo, d_m.cout), dtype=torch.bfloat16, device=x.device)
    dys, dym = dys.contiguous(), dym.contiguous()
    sync = getattr(lv_s.weights.graph, 'grad_sync', None)
    ready = (lambda lv: (lambda: sync.notify_layer_grad_ready(lv.weights))) if sync is not None else (lambda lv: None)
    # the strided conv: dW from x as it lies; dX on its own grid only (= the dgrad of a stride-1 1x1 conv over that grid)
    dxs = ops.conv_bwd_grid(d_s, x, dys, lv_s.hwio, lv_s.weights.grad.view(-1), on_dw_ready=ready(lv_s))
    dx = ops.conv_bwd(d_m, x, dym, lv_m.hwio, lv_m.weights.grad.view(-1), need_dx=True, addend=dxs,
                      addend_sub=(d_s.stride_h, d_s.stride_w), on_dw_ready=ready(lv_m))
    return dx, None, None, None, None, None


def conv_pair(conv_sub, conv_main, x, bn_stats=False):
  """(conv_sub(x), conv_main(x)) for a tensor read by exactly these two convs, conv_sub a strided 1x1 conv (see
  _MaskedConvPairFn); any other pair of layers -- and fp32 activations -- take conv_sub.fork(x) + conv_main(alias)."""
  n, h, w, c = x.shape
  ok = (x.dtype == torch.bfloat16 and x.requires_grad and conv_sub.need_input_grad and conv_main.need_input_grad
        and conv_sub.kh == conv_sub.kw == 1 and max(conv_sub.strides) > 1
        and conv_sub.cin % 8 == 0 and conv_sub.units % 8 == 0 and conv_main.units % 8 == 0
        and _CONV_PAIR and not (_BN_FUSE_BWD and getattr(x, 'bn_ctx', None) is not None))
  d_s = conv_sub.desc_for(n, h, w) if ok else None
  if ok and (d_s.pad_top or d_s.pad_left or d_s.ho != -(-h // d_s.stride_h) or d_s.wo != -(-w // d_s.stride_w)):
    ok = False
  if not ok:
    ys, alias = conv_sub.fork(x, bn_stats)
    return ys, conv_main(alias, bn_stats)
  if c != conv_sub.cin or c != conv_main.cin:
    raise ValueError('expected [N,H,W,%d], got %s' % (conv_sub.cin, tuple(x.shape)))
  conv_sub.graph.refresh_shadows()
  d_m = conv_main.desc_for(n, h, w)
  if not bn_stats:
    return _MaskedConvPairFn.apply(x.contiguous(), conv_sub.vars, d_s, conv_main.vars, d_m, False)
  ys, ym, ps, pm = _MaskedConvPairFn.apply(x.contiguous(), conv_sub.vars, d_s, conv_main.vars, d_m, True)
  if ps.numel():
    ys.bn_partials = ps
  if pm.numel():
    ym.bn_partials = pm
  return ys, ym


class _Layer:
  """Common part of MaskedConv2d / MaskedDense."""

  def __init__(self, graph, scope, shape, sparsity_technique, weight_decay,
               kernel_initializer):
    if sparsity_technique not in ('threshold', 'baseline'):
      raise ValueError(
          'Unsupported sparsity technique {}'.format(sparsity_technique))
    self.graph = graph
    self.scope = scope
    self.masked = sparsity_technique == 'threshold'
    init = kernel_initializer(shape) if kernel_initializer is not None else \
        variance_scaling_initializer()(shape)
    self.vars = graph.add_masked_layer(scope, shape, self.masked, weight_decay,
                                       init)
    self.bias = None

  @property
  def weights(self):
    return self.vars.weights

  @property
  def mask(self):
    return self.vars.mask


class MaskedConv2d(_Layer):
  """masked_conv2d / tf.layers.conv2d twin (pruning_layers.py:139-169)."""

  def __init__(self, graph, scope, cin, units, kernel_size, strides=(1, 1),
               padding='SAME', sparsity_technique='baseline', weight_decay=0.0,
               kernel_initializer=None, need_input_grad=True):
    kh, kw = kernel_size
    super().__init__(graph, scope, (kh, kw, cin, units), sparsity_technique,
                     weight_decay, kernel_initializer)
    if padding not in ('SAME', 'VALID'):
      raise ValueError('padding must be SAME or VALID')
    self.cin, self.units, self.kh, self.kw = cin, units, kh, kw
    self.strides = tuple(strides)
    self.padding = padding
    self.need_input_grad = need_input_grad
    self._descs = {}

  def desc_for(self, n, h, w):
    key = (n, h, w)
    d = self._descs.get(key)
    if d is None:
      sh, sw = self.strides
      if self.padding == 'SAME':
        ho, pt = _same_pad(h, self.kh, sh)
        wo, pl = _same_pad(w, self.kw, sw)
      else:
        ho, pt = _valid_out(h, self.kh, sh), 0
        wo, pl = _valid_out(w, self.kw, sw), 0
      d = ops.conv_desc(n, h, w, self.cin, self.units, self.kh, self.kw,
                        (sh, sw), pt, pl, ho, wo)
      self._descs[key] = d
    return d

  def __call__(self, x, bn_stats=False):
    """``bn_stats``: also leave the batch-norm partial sums of the output on the
    returned tensor (attribute ``bn_partials``) for the BatchNorm that follows,
    which then skips its statistics pass (rigl_masked_conv2d_fwd_stats)."""
    if x.dim() != 4:
      raise ValueError('Rank not supported {}'.format(x.dim()))
    if x.shape[-1] != self.cin:
      raise ValueError('expected %d input channels, got %d' %
                       (self.cin, x.shape[-1]))
    self.graph.refresh_shadows()
    n, h, w, _ = x.shape
    d = self.desc_for(n, h, w)
    need_dx = self.need_input_grad and x.requires_grad
    holder = _bn_source(x) if need_dx else None
    pending = getattr(x, 'bn_pending', None)   # the batch norm in front left its apply pass to this conv (takes_bn_input)
    if pending is not None and (pending.x is None or not x.is_contiguous()):
      raise RuntimeError('a deferred batch-norm output must reach its consumer conv as it was returned')
    if not x.requires_grad:
      x = x.detach().requires_grad_(True)  # keep the node so wgrad runs
    if not bn_stats:
      return _MaskedConvFn.apply(x.contiguous(), self.vars, d, need_dx, False, holder, pending)
    y, part = _MaskedConvFn.apply(x.contiguous(), self.vars, d, need_dx, True, holder, pending)
    if part.numel():
      y.bn_partials = part
    return y

  def takes_bn_input(self, x):
    """Does this conv's forward take relu(bn(.)) of its input on its operand load (rigl_masked_conv2d_fwd_bnrelu)?  ``x`` = the
    batch norm's input (the shape of this conv's input)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] == self.cin):
      return False
    n, h, w, _ = x.shape
    return ops.conv_fwd_takes_bn_input(self.desc_for(n, h, w))

  def takes_masked_addend(self, x):
    """Can the gradient of ``fork(x)``'s alias be handed to this conv unmasked with a 1-bit mask (rigl_masked_conv2d_bwd_masked)?
    Only then may the alias' other consumer (relu(bn3 + alias)) skip writing the masked gradient."""
    if not (self.need_input_grad and x.requires_grad and x.dtype == torch.bfloat16 and x.dim() == 4) or _BN_FUSE_BWD:
      return False
    n, h, w, _ = x.shape
    return ops.conv_bwd_takes_masked_addend(self.desc_for(n, h, w))

  def fork(self, x, bn_stats=False):
    """Returns (conv(x), x'): use x' for x's other consumer and its gradient
    is accumulated inside this conv's dgrad kernel (see _MaskedConvForkFn)."""
    if x.dim() != 4 or x.shape[-1] != self.cin:
      raise ValueError('expected [N,H,W,%d], got %s' % (self.cin, tuple(x.shape)))
    if not (self.need_input_grad and x.requires_grad):
      return self(x, bn_stats), x
    self.graph.refresh_shadows()
    n, h, w, _ = x.shape
    holder = _bn_source(x)
    if not bn_stats:
      return _MaskedConvForkFn.apply(x.contiguous(), self.vars, self.desc_for(n, h, w), False, holder)
    y, alias, part = _MaskedConvForkFn.apply(x.contiguous(), self.vars, self.desc_for(n, h, w), True, holder)
    if part.numel():
      y.bn_partials = part
    return y, alias


class MaskedDense(_Layer):
  """masked_fully_connected / tf.layers.dense twin (pruning_layers.py:222-245):
  y = x @ (mask*W) + b, weights [in, out]; runs as a 1x1 conv."""

  def __init__(self, graph, scope, n_in, units, use_bias=True,
               sparsity_technique='baseline', weight_decay=0.0,
               kernel_initializer=None, activation=None):
    super().__init__(graph, scope, (n_in, units), sparsity_technique,
                     weight_decay, kernel_initializer)
    self.n_in, self.units = n_in, units
    self.activation = activation
    self._descs = {}
    if use_bias:
      self.bias = graph.add_variable(scope + '/biases', (units,), V.KIND_OTHER, 0.0)

  def __call__(self, x):
    if x.shape[-1] != self.n_in:
      raise ValueError('expected %d inputs, got %d' % (self.n_in, x.shape[-1]))
    self.graph.refresh_shadows()
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 1, 1, self.n_in)
    b = x2.shape[0]
    d = self._descs.get(b)
    if d is None:
      d = ops.conv_desc(b, 1, 1, self.n_in, self.units, 1, 1, 1, 0, 0, 1, 1)
      self._descs[b] = d
    need_dx = x2.requires_grad
    if not x2.requires_grad:
      x2 = x2.detach().requires_grad_(True)
    y = _MaskedConvFn.apply(x2.contiguous(), self.vars, d, need_dx)
    y = y.reshape(*lead, self.units)
    if self.bias is not None:
      y = y + bias_tensor(self.bias).to(y.dtype)
    if self.activation is not None:
      y = self.activation(y)
    return y


def bias_tensor(var):
  """An autograd leaf aliasing the variable's storage whose .grad IS the
  variable's slice of the gradient arena (autograd accumulates in place).
  The variable's own ``data`` tensor never requires grad, so the arenas stay
  outside any autograd graph; the kernels update them in place."""
  leaf = getattr(var, '_leaf', None)
  if leaf is None or leaf.data_ptr() != var.data.data_ptr():
    leaf = var.data.detach().requires_grad_(True)
    var._leaf = leaf
  if leaf.grad is None or leaf.grad.data_ptr() != var.grad.data_ptr():
    leaf.grad = var.grad
  return leaf


# ----------------------------------------------------------------------------
# initializers (host side, NumPy; shapes are HWIO / [in, out])
# ----------------------------------------------------------------------------
_init_rng = np.random.RandomState(0)


def set_init_seed(seed):
  global _init_rng
  _init_rng = np.random.RandomState(seed)


def variance_scaling_initializer(scale=1.0):
  """tf.variance_scaling_initializer(scale): fan_in, truncated normal
  (resnet_model.py:283)."""

  def init(shape):
    fan_in = int(np.prod(shape[:-1]))
    std = math.sqrt(scale / max(1.0, fan_in)) / .87962566103423978
    v = _init_rng.randn(*shape)
    bad = np.abs(v) > 2
    while bad.any():
      v[bad] = _init_rng.randn(int(bad.sum()))
      bad = np.abs(v) > 2
    return (v * std).astype(np.float32)

  return init


def random_normal_initializer(stddev=0.01):
  return lambda shape: (_init_rng.randn(*shape) * stddev).astype(np.float32)


# ----------------------------------------------------------------------------
# the reference's functional API
# ----------------------------------------------------------------------------
def _regularizer_scale(reg):
  """kernel_regularizer may be a float (l2 scale) or an object with .scale."""
  if reg is None:
    return 0.0
  if isinstance(reg, (int, float)):
    return float(reg)
  return float(getattr(reg, 'scale'))


class l2_regularizer:  # pylint: disable=invalid-name
  """contrib_layers.l2_regularizer(scale): scale * sum(w^2) / 2; its gradient
  (scale * w) is applied inside the fused update kernel (K3)."""

  def __init__(self, scale):
    self.scale = float(scale)


def sparse_conv2d(x, units, kernel_size, activation=None, use_bias=False,
                  kernel_initializer=None, kernel_regularizer=None,
                  bias_initializer=None, biases_regularizer=None,
                  sparsity_technique='baseline', normalizer_fn=None,
                  strides=(1, 1), padding='SAME', data_format='channels_last',
                  name=None, graph=None):
  """rigl/imagenet_resnet/pruning_layers.py:72-172.  Variables are created on
  the first call for a scope and reused afterwards (tf.variable_scope)."""
  del bias_initializer, biases_regularizer
  if data_format == 'channels_last':
    pass
  elif data_format == 'channels_first':
    raise ValueError('channels_first is not supported by the NHWC kernels')
  else:
    raise ValueError('Not a valid channel string:', data_format)
  if x.dim() != 4:
    raise ValueError('Rank not supported {}'.format(x.dim()))
  if sparsity_technique not in ('threshold', 'baseline'):
    raise ValueError(
        'Unsupported sparsity technique {}'.format(sparsity_technique))
  if use_bias:
    raise ValueError('use_bias=True is not used by the reference models')
  g = graph or V.get_default_graph()
  key = name
  layer = g.modules.get(key) if key is not None else None
  if layer is None:
    scope = g.unique_scope(name, 'Conv')
    layer = MaskedConv2d(g, scope, x.shape[-1], units, tuple(kernel_size),
                         tuple(strides), padding, sparsity_technique,
                         _regularizer_scale(kernel_regularizer),
                         kernel_initializer)
    g.modules[scope] = layer
  y = layer(x)
  if normalizer_fn is not None:
    y = normalizer_fn(y)
  if activation is not None:
    y = activation(y)
  return y


def sparse_fully_connected(x, units, activation=None, use_bias=True,
                           kernel_initializer=None, kernel_regularizer=None,
                           bias_initializer=None, biases_regularizer=None,
                           sparsity_technique='baseline', name=None,
                           graph=None):
  """rigl/imagenet_resnet/pruning_layers.py:175-248."""
  del bias_initializer, biases_regularizer
  if sparsity_technique not in ('threshold', 'baseline'):
    raise ValueError(
        'Unsupported sparsity technique {}'.format(sparsity_technique))
  g = graph or V.get_default_graph()
  layer = g.modules.get(name) if name is not None else None
  if layer is None:
    scope = g.unique_scope(name, 'Dense')
    layer = MaskedDense(g, scope, x.shape[-1], units, use_bias,
                        sparsity_technique,
                        _regularizer_scale(kernel_regularizer),
                        kernel_initializer, activation)
    g.modules[scope] = layer
  return layer(x)
