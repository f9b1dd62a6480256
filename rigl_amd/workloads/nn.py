"""Glue ops between the masked convolutions: batch-norm, ReLU, pooling, loss.

These are NOT on the graded hot path (SURVEY 8a lists conv/fc fwd+bwd, the
mask update and the optimizer update); they are HBM-bound plumbing and run as
PyTorch-ROCm ops on NHWC bf16 tensors.  All tensors here are [N,H,W,C]
contiguous; PyTorch's NCHW-shaped channels_last view of the same memory is
used where an op wants NCHW.
"""
import torch
import torch.nn.functional as F

from rigl_amd import variables as V
from rigl_amd.pruning_layers import bias_tensor

BATCH_NORM_DECAY = 0.9      # resnet_model.py:37
BATCH_NORM_EPSILON = 1e-5   # resnet_model.py:38


def nchw_view(x):
  """[N,H,W,C] contiguous -> NCHW-shaped channels_last view (no copy)."""
  return x.permute(0, 3, 1, 2)


def nhwc_view(x):
  """NCHW-shaped channels_last tensor -> [N,H,W,C] contiguous view."""
  return x.permute(0, 2, 3, 1)


class BatchNorm:
  """tf.layers.batch_normalization(momentum=0.9, eps=1e-5, fused) over the
  channel axis (resnet_model.py:41-82).  gamma/beta live in the graph's
  BN/bias arena segment; moving statistics are plain buffers."""

  def __init__(self, graph, scope, channels, init_zero=False,
               decay=BATCH_NORM_DECAY, eps=BATCH_NORM_EPSILON):
    self.gamma = graph.add_variable(scope + '/gamma', (channels,), V.KIND_OTHER,
                                    0.0, init=None)
    if not init_zero:
      self.gamma.data.fill_(1.0)
    self.beta = graph.add_variable(scope + '/beta', (channels,), V.KIND_OTHER)
    self.moving_mean = torch.zeros(channels, device=graph.device)
    self.moving_variance = torch.ones(channels, device=graph.device)
    self.decay, self.eps = decay, eps

  def __call__(self, x, is_training=True, relu=False):
    y = F.batch_norm(nchw_view(x), self.moving_mean, self.moving_variance,
                     bias_tensor(self.gamma), bias_tensor(self.beta),
                     is_training, 1.0 - self.decay, self.eps)
    if relu:
      y = F.relu(y, inplace=True)
    return nhwc_view(y)


def max_pool_3x3_s2_same(x):
  """tf.layers.max_pooling2d(pool_size=3, strides=2, padding='SAME')
  (resnet_model.py:637-644): TF pads (0,1) on even inputs, i.e. only at the
  bottom / right -- not torchvision's symmetric pad 1."""
  xn = nchw_view(x)
  h, w = xn.shape[2], xn.shape[3]
  ph = max((-(-h // 2) - 1) * 2 + 3 - h, 0)
  pw = max((-(-w // 2) - 1) * 2 + 3 - w, 0)
  xn = F.pad(xn, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2),
             value=float('-inf'))
  return nhwc_view(F.max_pool2d(xn, 3, 2))


def global_avg_pool(x):
  """average_pooling2d over the whole map + reshape (resnet_model.py:701-712)."""
  return x.float().mean(dim=(1, 2)).to(x.dtype)


def softmax_cross_entropy(logits, labels, label_smoothing=0.0):
  """tf.losses.softmax_cross_entropy(onehot, logits, label_smoothing): targets
  onehot*(1-eps) + eps/K, mean over the batch (imagenet_train_eval.py:578-584)."""
  return F.cross_entropy(logits.float(), labels,
                         label_smoothing=label_smoothing)
