"""Glue ops between the masked convolutions: batch-norm, ReLU, pooling, loss.

These are NOT on the graded hot path (SURVEY 8a lists conv/fc fwd+bwd, the
mask update and the optimizer update); they are HBM-bound plumbing and run as
PyTorch-ROCm ops on NHWC bf16 tensors.  All tensors here are [N,H,W,C]
contiguous; PyTorch's NCHW-shaped channels_last view of the same memory is
used where an op wants NCHW.
"""
import os

import torch
import torch.nn.functional as F

from rigl_amd import variables as V
from rigl_amd.pruning_layers import bias_tensor

BATCH_NORM_DECAY = 0.9      # resnet_model.py:37
BATCH_NORM_EPSILON = 1e-5   # resnet_model.py:38
_DW_STATS = os.environ.get('RIGL_DW_STATS', '1') != '0'   # depthwise forward leaves the next batch norm's statistics


def activation_dtype(precision=None):
  """The dtype of the activations for the reference's ``--precision`` flag (imagenet_train_eval.py:56-59): 'bfloat16'
  (what the MFMA kernels and every measured number use) or 'float32' (the fp32 validation kernels of K1, stock ops
  for the glue).  ``None``: bfloat16 unless knob "k1_fp32" (RIGL_K1_FP32=1) is set."""
  if precision is None:
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    precision = 'float32' if ops.tune_get('k1_fp32', 0) == 1 else 'bfloat16'
  if precision not in ('bfloat16', 'float32'):
    raise ValueError('precision must be bfloat16 or float32, got %r' % (precision,))
  return torch.float32 if precision == 'float32' else torch.bfloat16


def nchw_view(x):
  """[N,H,W,C] contiguous -> NCHW-shaped channels_last view (no copy)."""
  return x.permute(0, 3, 1, 2)


def nhwc_view(x):
  """NCHW-shaped channels_last tensor -> [N,H,W,C] contiguous view."""
  return x.permute(0, 2, 3, 1)


class _BnBwdHolder:
  """What the consumer of a batch norm's output needs to take that batch norm's backward reductions in its own dgrad
  epilogue (pruning_layers._bn_source), and where it leaves them for the batch norm's backward."""
  __slots__ = ('x', 'saved', 'bits', 'relu', 'attached', 'partials', 'dx_ptr')

  def __init__(self, relu):
    self.x = self.saved = self.bits = self.partials = self.dx_ptr = None
    self.relu = relu
    self.attached = 0


_LAZY_RES_GRAD = os.environ.get('RIGL_LAZY_RES_GRAD', '1') != '0'
# relu(bn(x)) applied on the operand load of the conv that consumes it (ops.conv_fwd_bnrelu): built and bit-identical, measured
# SLOWER than the separate apply pass (profiles/r6/README.md: the activated tensor still has to be written for the backward), so off
_BN_ON_LOAD = os.environ.get('RIGL_BN_ON_LOAD', '0') == '1'


class _FusedBNFn(torch.autograd.Function):
  """y = relu?(bn(x) (+ residual)) through rigl_bn_fwd / rigl_bn_bwd.  The
  parameter gradients go straight into the gradient arena (overwrite), like
  the conv kernels' dW; autograd only routes dx (and the residual's grad)."""

  @staticmethod
  def forward(ctx, x, residual, bn, relu, partials=None, holder=None, lazy_res_grad=False, defer=False):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x = x.contiguous()
    res = residual.contiguous() if residual is not None else None
    ctx.bn, ctx.relu, ctx.has_res = bn, relu, res is not None
    ctx.holder = holder
    ctx.lazy_res_grad = bool(lazy_res_grad) and relu and res is not None
    if defer:
      # statistics only: the consumer conv applies the batch norm + ReLU on its operand load and FILLS y as a side output
      # (MaskedConv2d picks x and saved up from the holder: y.bn_pending)
      saved = ops.bn_statistics(x, bn.gamma.data, bn.beta.data, bn.moving_mean, bn.moving_variance, 1.0 - bn.decay, bn.eps,
                                partials=partials)
      ctx.save_for_backward(x, saved)
      holder.x, holder.saved, holder.bits = x, saved, None
      return torch.empty_like(x)
    # the ReLU mask of relu(bn + residual) is kept as 1 bit per element (the backward would
    # otherwise re-read the whole output twice); without a residual it is recomputed from x
    if relu and res is not None:
      y, saved, bits = ops.bn_fwd(x, bn.gamma.data, bn.beta.data, bn.moving_mean,
                                  bn.moving_variance, 1.0 - bn.decay, bn.eps, relu, res,
                                  partials=partials, want_relu_bits=True)
      ctx.save_for_backward(x, saved, bits)
    else:
      bits = None
      y, saved = ops.bn_fwd(x, bn.gamma.data, bn.beta.data, bn.moving_mean,
                            bn.moving_variance, 1.0 - bn.decay, bn.eps, relu, res,
                            partials=partials)
      ctx.save_for_backward(x, saved)
    if holder is not None:
      holder.x, holder.saved, holder.bits = x, saved, bits
    return y

  @staticmethod
  def backward(ctx, dy):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    bn = ctx.bn
    if ctx.relu and ctx.has_res:
      x, saved, bits = ctx.saved_tensors
    else:
      (x, saved), bits = ctx.saved_tensors, None
    dy = dy.contiguous()
    # sum dz / sum dz * xhat already taken by the kernel that produced dy (the sole consumer's dgrad epilogue)?
    h, part = ctx.holder, None
    if h is not None:
      if h.partials is not None and h.dx_ptr == dy.data_ptr():
        part = h.partials
      h.partials = h.x = h.saved = h.bits = None
    if ctx.lazy_res_grad and bits is not None and ctx.needs_input_grad[1]:
      # the residual's consumer (the block's first conv, pruning_layers._MaskedConvForkFn) masks on the fly: its gradient is
      # dy where the ReLU was on, so dy itself travels with the ReLU bits and the masked copy is never written
      dx, _ = ops.bn_bwd(x, None, dy, bn.gamma.data, saved, ctx.relu, bn.gamma.grad, bn.beta.grad, want_dres=False,
                         relu_bits=bits, partials=part)
      ops.LAZY_ADDEND_BITS[dy.data_ptr()] = (bits, dy)
      return dx, dy, None, None, None, None, None, None
    dx, dres = ops.bn_bwd(x, None, dy, bn.gamma.data, saved, ctx.relu,
                          bn.gamma.grad, bn.beta.grad,
                          want_dres=ctx.has_res and ctx.needs_input_grad[1],
                          relu_bits=bits, partials=part)
    return dx, dres, None, None, None, None, None, None


class _BnAddBnFn(torch.autograd.Function):
  """relu(bn(x) + bn2(x2)) as one node (rigl_bn_add_bn_fwd / _bwd): the normalised shortcut and the relu-masked
  gradient between the two batch norms are never written.  Bit-identical to bn(x, residual=bn2(x2))."""

  @staticmethod
  def forward(ctx, x, x2, bn, bn2, partials, partials2):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x, x2 = x.contiguous(), x2.contiguous()
    saved2 = ops.bn_statistics(x2, bn2.gamma.data, bn2.beta.data, bn2.moving_mean, bn2.moving_variance,
                               1.0 - bn2.decay, bn2.eps, partials=partials2)
    y, saved, bits = ops.bn_add_bn_fwd(x, x2, saved2, bn.gamma.data, bn.beta.data, bn.moving_mean, bn.moving_variance,
                                       1.0 - bn.decay, bn.eps, True, partials=partials)
    ctx.bn, ctx.bn2 = bn, bn2
    ctx.save_for_backward(x, x2, saved, saved2, bits)
    return y

  @staticmethod
  def backward(ctx, dy):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x, x2, saved, saved2, bits = ctx.saved_tensors
    bn, bn2 = ctx.bn, ctx.bn2
    dx, dx2 = ops.bn_add_bn_bwd(x, x2, bits, dy.contiguous(), bn.gamma.data, saved, bn2.gamma.data, saved2,
                                bn.gamma.grad, bn.beta.grad, bn2.gamma.grad, bn2.beta.grad)
    return dx, dx2, None, None, None, None


_BN_PAIR_FUSED = os.environ.get('RIGL_BN_PAIR', '1') != '0'


def bn_add_bn_relu(bn, x, bn2, x2, is_training=True):
  """relu(bn(x) + bn2(x2)): a residual block whose shortcut is projection conv + batch norm
  (resnet_model.py:456-501)."""
  c = x.shape[-1]
  if (_BN_PAIR_FUSED and bn.fused and bn2.fused and is_training and x.is_cuda and x.dtype == torch.bfloat16
      and x2.dtype == torch.bfloat16 and x2.shape == x.shape and c % 8 == 0 and c <= 3968):
    px, px2 = getattr(x, 'bn_partials', None), getattr(x2, 'bn_partials', None)
    if not x.requires_grad:
      x = x.detach().requires_grad_(True)
    if not x2.requires_grad:
      x2 = x2.detach().requires_grad_(True)
    return _BnAddBnFn.apply(x, x2, bn, bn2, px, px2)
  return bn(x, is_training, relu=True, residual=bn2(x2, is_training, relu=False))


class BatchNorm:
  """tf.layers.batch_normalization(momentum=0.9, eps=1e-5, fused) over the
  channel axis (resnet_model.py:41-82), optionally fused with the residual add
  and the ReLU that follow it.  gamma/beta live in the graph's BN/bias arena
  segment; moving statistics are plain buffers.  Training mode with
  channels % 8 == 0 runs the fused HIP kernels (bn.hip); anything else falls
  back to the stock PyTorch ops (this is glue, not the graded path)."""

  def __init__(self, graph, scope, channels, init_zero=False,
               decay=BATCH_NORM_DECAY, eps=BATCH_NORM_EPSILON, fused=True):
    self.gamma = graph.add_variable(scope + '/gamma', (channels,), V.KIND_OTHER,
                                    0.0, init=None)
    if not init_zero:
      self.gamma.data.fill_(1.0)
    self.beta = graph.add_variable(scope + '/beta', (channels,), V.KIND_OTHER)
    self.moving_mean = torch.zeros(channels, device=graph.device)
    self.moving_variance = torch.ones(channels, device=graph.device)
    self.decay, self.eps = decay, eps
    self.channels = channels
    self.fused = fused
    self.scope = scope
    graph.modules[scope] = self          # creation order = TF's batch_normalization_<k> numbering

  def __call__(self, x, is_training=True, relu=False, residual=None, lazy_res_grad=False, consumer=None):
    """``lazy_res_grad``: the residual's ONLY other consumer is a masked conv whose backward takes its addend unmasked with
    the ReLU bits (conv.takes_masked_addend): the backward then hands the output gradient itself to it.
    ``consumer``: the masked conv that is the ONLY reader of the output and is called on it next; where its forward takes the
    batch norm on its operand load (conv.takes_bn_input) only the statistics are finalised here and the conv fills the output."""
    if (self.fused and is_training and x.is_cuda and self.channels % 8 == 0
        and x.dtype == torch.bfloat16):
      partials = getattr(x, 'bn_partials', None)   # left by the producing conv's epilogue
      if not x.requires_grad:
        x = x.detach().requires_grad_(True)
      holder = _BnBwdHolder(relu)
      defer = bool(_BN_ON_LOAD and consumer is not None and relu and residual is None and consumer.takes_bn_input(x))
      y = _FusedBNFn.apply(x, residual, self, relu, partials, holder, lazy_res_grad and _LAZY_RES_GRAD, defer)
      y.bn_ctx = holder                            # a masked conv that is this tensor's only consumer picks it up
      if defer:
        y.bn_pending = holder
      return y
    y = F.batch_norm(nchw_view(x), self.moving_mean, self.moving_variance,
                     bias_tensor(self.gamma), bias_tensor(self.beta),
                     is_training, 1.0 - self.decay, self.eps)
    y = nhwc_view(y)
    if residual is not None:
      y = y + residual
    if relu:
      y = F.relu(y)
    return y


class _DepthwiseFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, layer, desc, want_stats=False):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x = x.contiguous()
    ctx.layer, ctx.desc = layer, desc
    ctx.save_for_backward(x)
    if not want_stats:
      return ops.depthwise_fwd(desc, x, layer.weights.data.view(-1))
    y, part = ops.depthwise_fwd(desc, x, layer.weights.data.view(-1), stats=True)
    if part is None:
      part = torch.empty(0, device=x.device)
    ctx.mark_non_differentiable(part)
    ctx.set_materialize_grads(False)
    return y, part

  @staticmethod
  def backward(ctx, dy, _dpart=None):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    (x,) = ctx.saved_tensors
    dy = dy.contiguous()
    w = ctx.layer.weights
    ops.depthwise_wgrad(ctx.desc, x, dy, w.grad.view(-1))          # dense fp32, into the gradient arena
    return ops.depthwise_dgrad(ctx.desc, dy, w.data.view(-1)), None, None, None


class DepthwiseConv2d:
  """depthwise_conv2d_fixed_padding (mobilenetv1_model.py:43-92): dense
  depthwise kxk, no bias, no regulariser, NOT masked; stride > 1 uses explicit
  fixed padding + VALID, stride 1 uses SAME."""

  def __init__(self, graph, scope, channels, kernel_size=3, stride=1):
    from rigl_amd import pruning_layers as PL  # pylint: disable=import-outside-toplevel
    self.k, self.stride, self.channels = kernel_size, stride, channels
    init = PL.variance_scaling_initializer()((kernel_size, kernel_size, channels, 1))
    self.weights = graph.add_variable(scope + '/depthwise_weights', (kernel_size, kernel_size, channels, 1),
                                      V.KIND_OTHER, 0.0, init)
    self._descs = {}

  def __call__(self, x, bn_stats=False):
    """``bn_stats``: leave the batch-norm partial sums of the output on the returned tensor (``bn_partials``) for the
    BatchNorm that follows, which then skips its statistics pass (rigl_depthwise_conv2d_fwd_stats)."""
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    n, h, w, c = x.shape
    d = self._descs.get((n, h, w))
    if d is None:
      pad = (self.k - 1) // 2
      ho, wo = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
      d = ops.conv_desc(n, h, w, c, c, self.k, self.k, self.stride, pad, pad, ho, wo)
      self._descs[(n, h, w)] = d
    if not x.requires_grad:
      x = x.detach().requires_grad_(True)
    if not (bn_stats and _DW_STATS):
      return _DepthwiseFn.apply(x, self, d)
    y, part = _DepthwiseFn.apply(x, self, d, True)
    if part.numel():
      y.bn_partials = part
    return y


class _MaxPoolFn(torch.autograd.Function):
  """NHWC bf16 max pooling through rigl_maxpool_fwd / rigl_maxpool_bwd."""

  @staticmethod
  def forward(ctx, x, desc):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    y, arg = ops.maxpool_fwd(desc, x.contiguous())
    ctx.desc = desc
    ctx.save_for_backward(arg)
    return y

  @staticmethod
  def backward(ctx, dy):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    (arg,) = ctx.saved_tensors
    return ops.maxpool_bwd(ctx.desc, dy.contiguous(), arg), None


class _BnReluMaxPoolFn(torch.autograd.Function):
  """maxpool_3x3_s2_same(relu(bn(x))) as one node: the activated tensor is never written, the backward's two
  batch-norm passes gather the pooling gradient on the fly (rigl_bn_relu_maxpool_fwd / _bwd)."""

  @staticmethod
  def forward(ctx, x, bn, desc, partials=None):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x = x.contiguous()
    y, arg, saved = ops.bn_relu_maxpool_fwd(desc, x, bn.gamma.data, bn.beta.data, bn.moving_mean, bn.moving_variance,
                                            1.0 - bn.decay, bn.eps, partials=partials)
    ctx.bn, ctx.desc = bn, desc
    ctx.save_for_backward(x, arg, saved)
    return y

  @staticmethod
  def backward(ctx, dy):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    x, arg, saved = ctx.saved_tensors
    bn = ctx.bn
    dx = ops.bn_relu_maxpool_bwd(ctx.desc, x, dy.contiguous(), arg, bn.gamma.data, saved, bn.gamma.grad, bn.beta.grad)
    return dx, None, None, None


_STEM_TAIL_FUSED = os.environ.get('RIGL_STEM_TAIL', '1') != '0'


def _pool_desc(x):
  from rigl_amd import ops  # pylint: disable=import-outside-toplevel
  n, h, w, c = x.shape
  key = (n, h, w, c)
  d = _POOL_DESCS.get(key)
  if d is None:
    ho, wo = -(-h // 2), -(-w // 2)
    ph = max((ho - 1) * 2 + 3 - h, 0)
    pw = max((wo - 1) * 2 + 3 - w, 0)
    d = ops.conv_desc(n, h, w, c, c, 3, 3, 2, ph // 2, pw // 2, ho, wo)
    _POOL_DESCS[key] = d
  return d


def bn_relu_max_pool_3x3_s2_same(bn, x, is_training=True):
  """batch_norm_relu followed by tf.layers.max_pooling2d(3, 2, 'SAME') (resnet_model.py:631-644)."""
  c = x.shape[-1]
  if (_STEM_TAIL_FUSED and bn.fused and is_training and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
      and c % 8 == 0 and c <= 256 and 64 % (c // 4) == 0 and x.numel() // 4 < (1 << 31)):
    partials = getattr(x, 'bn_partials', None)
    if not x.requires_grad:
      x = x.detach().requires_grad_(True)
    return _BnReluMaxPoolFn.apply(x, bn, _pool_desc(x), partials)
  return max_pool_3x3_s2_same(bn(x, is_training, relu=True))


_POOL_DESCS = {}


def max_pool_3x3_s2_same(x):
  """tf.layers.max_pooling2d(pool_size=3, strides=2, padding='SAME')
  (resnet_model.py:637-644): TF pads (0,1) on even inputs, i.e. only at the
  bottom / right -- not torchvision's symmetric pad 1."""
  if x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] % 8 == 0:
    return _MaxPoolFn.apply(x, _pool_desc(x))
  # TF SAME on an even size pads only bottom/right with -inf; windows start at
  # 0, 2, 4, ... -- exactly max_pool2d(3, 2, padding=0, ceil_mode=True), with no
  # padded copy of the 205 MB stem activation.
  xn = nchw_view(x)
  h, w = xn.shape[2], xn.shape[3]
  if h % 2 == 0 and w % 2 == 0:
    return nhwc_view(F.max_pool2d(xn, 3, 2, padding=0, ceil_mode=True))
  ph = max((-(-h // 2) - 1) * 2 + 3 - h, 0)
  pw = max((-(-w // 2) - 1) * 2 + 3 - w, 0)
  xn = F.pad(xn, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2),
             value=float('-inf'))
  return nhwc_view(F.max_pool2d(xn, 3, 2))


# RIGL_HEAD_TORCH=1: the PyTorch formulation of the classifier head (comparison runs)
_HEAD_KERNELS = os.environ.get('RIGL_HEAD_TORCH', '0') != '1'


class _GlobalAvgPoolFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    ctx.hw = (x.shape[1], x.shape[2])
    return ops.global_avgpool_fwd(x.contiguous())

  @staticmethod
  def backward(ctx, dy):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    return ops.global_avgpool_bwd(dy.contiguous(), *ctx.hw)


def global_avg_pool(x):
  """average_pooling2d over the whole map + reshape (resnet_model.py:701-712)."""
  if _HEAD_KERNELS and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 2 == 0:
    return _GlobalAvgPoolFn.apply(x)          # one kernel each way instead of cast + reduce + cast (+ div + cast)
  return x.float().mean(dim=(1, 2)).to(x.dtype)


class _SoftmaxXentFn(torch.autograd.Function):
  """Mean cross entropy with label smoothing; the kernel leaves d(mean loss)/d(logits) next to the row losses."""

  @staticmethod
  def forward(ctx, logits, labels, label_smoothing):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    rows, dz = ops.softmax_xent(logits.contiguous(), labels.reshape(-1).contiguous(), label_smoothing)
    ctx.save_for_backward(dz)
    return rows.mean()

  @staticmethod
  def backward(ctx, g):
    dz, = ctx.saved_tensors
    return (dz * g.to(dz.dtype)), None, None


def softmax_cross_entropy(logits, labels, label_smoothing=0.0):
  """tf.losses.softmax_cross_entropy(onehot, logits, label_smoothing): targets
  onehot*(1-eps) + eps/K, mean over the batch (imagenet_train_eval.py:578-584).
  Written out with log_softmax + gather: F.cross_entropy(label_smoothing=...)
  synchronises the device on ROCm (measured, tools/sync_probe.py: the host blocks
  until every queued kernel has finished), which drained the launch queue once
  per step."""
  if (_HEAD_KERNELS and logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2 and
      labels.dtype == torch.int64):
    return _SoftmaxXentFn.apply(logits, labels, float(label_smoothing))
  logp = F.log_softmax(logits.float(), dim=-1)
  nll = -logp.gather(1, labels.reshape(-1, 1)).squeeze(1)
  if not label_smoothing:
    return nll.mean()
  k = logits.shape[-1]
  return ((1.0 - label_smoothing) * nll - (label_smoothing / k) * logp.sum(dim=1)).mean()
