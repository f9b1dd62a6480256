"""ResNet-50 v1.5 workload built from the masked layers (the reference's
``resnet_v1_(50, ...)``, rigl/imagenet_resnet/resnet_model.py:734-805), with
synthetic ImageNet-shaped input.  Activations NHWC bf16, fp32 master weights.

Topology facts kept from the reference: stride on the 3x3
(v1.5); strided convs = fixed_padding + VALID, stride-1 convs = SAME
(:278-281); projection shortcut conv+BN in the first block of each group;
gamma = 0 on the last BN of every block (:498-499); stem 7x7/2 -> BN-ReLU ->
max-pool 3x3/2 'SAME'; global average pool; final_dense 2048->1000 with bias,
N(0, 0.01) init (:713); l2 (weight_decay) on every conv / dense kernel.
"""
import torch

from rigl_amd import pruning_layers as PL
from rigl_amd import variables as V
from rigl_amd.workloads import nn as gnn
from rigl_amd.workloads import shapes as WS


class _ConvFixedPadding:
  """conv2d_fixed_padding (resnet_model.py:234-303): for stride > 1 pad
  (k-1)//2 on every side explicitly and convolve 'VALID'; else 'SAME'."""

  def __init__(self, graph, spec, technique, weight_decay, need_input_grad=True):
    scope = WS.SCOPE + '/' + spec.end_point
    self.k, self.stride = spec.k, spec.stride
    self.conv = PL.MaskedConv2d(
        graph, scope, spec.cin, spec.cout, (spec.k, spec.k),
        (spec.stride, spec.stride), 'SAME', technique, weight_decay,
        PL.variance_scaling_initializer(), need_input_grad)
    graph.modules[scope] = self.conv
    if spec.stride > 1:
      self.conv.desc_for = self._desc_fixed      # explicit symmetric padding

  def _desc_fixed(self, n, h, w):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    c = self.conv
    key = (n, h, w)
    d = c._descs.get(key)
    if d is None:
      pad = (self.k - 1) // 2                     # pad_beg of fixed_padding (:98-100)
      ho = (h + (self.k - 1) - self.k) // self.stride + 1
      wo = (w + (self.k - 1) - self.k) // self.stride + 1
      d = ops.conv_desc(n, h, w, c.cin, c.units, self.k, self.k, self.stride,
                        pad, pad, ho, wo)
      c._descs[key] = d
    return d

  def __call__(self, x, bn_stats=True):
    return self.conv(x, bn_stats)

  def fork(self, x, bn_stats=True):
    return self.conv.fork(x, bn_stats)

  def takes_masked_addend(self, x):
    return self.conv.takes_masked_addend(x)

  def takes_bn_input(self, x):
    return self.conv.takes_bn_input(x)


class _Bottleneck:
  """bottleneck_block_ (resnet_model.py:396-501)."""

  def __init__(self, graph, convs, technique, weight_decay, tag):
    by_role = {c.role: c for c in convs}
    self.proj = self.proj_bn = None
    if 'proj' in by_role:
      self.proj = _ConvFixedPadding(graph, by_role['proj'], technique, weight_decay)
      self.proj_bn = gnn.BatchNorm(graph, tag + '/bn_proj', by_role['proj'].cout)
    self.c1 = _ConvFixedPadding(graph, by_role['c1'], technique, weight_decay)
    self.bn1 = gnn.BatchNorm(graph, tag + '/bn1', by_role['c1'].cout)
    self.c2 = _ConvFixedPadding(graph, by_role['c2'], technique, weight_decay)
    self.bn2 = gnn.BatchNorm(graph, tag + '/bn2', by_role['c2'].cout)
    self.c3 = _ConvFixedPadding(graph, by_role['c3'], technique, weight_decay)
    self.bn3 = gnn.BatchNorm(graph, tag + '/bn3', by_role['c3'].cout, init_zero=True)

  def __call__(self, x, is_training):
    # The block input feeds two consumers; the first one hands back an alias whose
    # gradient it accumulates in its own dgrad epilogue (no separate AddN pass).
    if self.proj is not None:
      # (a strided projection and conv1 as one autograd node: the projection's input gradient stays on its own grid)
      p, y = PL.conv_pair(self.proj.conv, self.c1.conv, x, bn_stats=True)
    else:
      lazy = self.c1.takes_masked_addend(x)      # (the shortcut's gradient then reaches conv1 unmasked + the ReLU bits)
      y, shortcut = self.c1.fork(x)
    y = self.bn1(y, is_training, relu=True)
    y = self.bn2(self.c2(y), is_training, relu=True, consumer=self.c3)   # (RIGL_BN_ON_LOAD=1: conv3 applies it on its operand load)
    if self.proj is not None:
      # relu(bn3(conv3) + bn_proj(projection)) in one piece: neither the normalised shortcut nor the masked gradient
      # between the two batch norms is written
      return gnn.bn_add_bn_relu(self.bn3, self.c3(y), self.proj_bn, p, is_training)
    return self.bn3(self.c3(y), is_training, relu=True, residual=shortcut, lazy_res_grad=lazy)   # relu(bn3 + shortcut)


class ResNet50:

  def __init__(self, graph=None, num_classes=1000, pruning_method='threshold',
               prune_first_layer=True, prune_last_layer=True, weight_decay=1e-4,
               seed=0):
    self.graph = graph or V.get_default_graph()
    g = self.graph
    PL.set_init_seed(seed)
    tech = pruning_method
    stem_spec = WS.ConvSpec('initial_conv', 7, 3, 64, 2, 'stem')
    self.stem = _ConvFixedPadding(g, stem_spec, tech if prune_first_layer else 'baseline',
                                  weight_decay, need_input_grad=False)
    self.stem_bn = gnn.BatchNorm(g, WS.SCOPE + '/initial_bn', 64)
    self.blocks = []
    for grp, n, convs in WS.resnet50_blocks():
      self.blocks.append(_Bottleneck(g, convs, tech, weight_decay,
                                     '%s/group%d_block%d' % (WS.SCOPE, grp, n)))
    self.fc = PL.MaskedDense(g, WS.SCOPE + '/final_dense', 2048, num_classes, True,
                             tech if prune_last_layer else 'baseline', weight_decay,
                             PL.random_normal_initializer(0.01))
    g.modules[WS.SCOPE + '/final_dense'] = self.fc
    g.finalize()

  def __call__(self, images, is_training=True):
    """images: [N,224,224,3] bf16 (already mean/std normalised)."""
    x = gnn.bn_relu_max_pool_3x3_s2_same(self.stem_bn, self.stem(images), is_training)
    for b in self.blocks:
      x = b(x, is_training)
    x = gnn.global_avg_pool(x)
    return self.fc(x)

  def loss(self, images, labels, label_smoothing=0.1, is_training=True):
    """Cross entropy with label smoothing (imagenet_train_eval.py:578-584).
    The l2 term's gradient is applied by the fused update kernel."""
    return gnn.softmax_cross_entropy(self(images, is_training), labels,
                                     label_smoothing)


def synthetic_batch(batch, device, seed=1234, image_size=224, num_classes=1000, precision=None):
  """N(0,1) images (already 'normalised'), uniform labels.  ``precision``: gnn.activation_dtype."""
  gen = torch.Generator(device=device).manual_seed(seed)
  images = torch.randn(batch, image_size, image_size, 3, generator=gen,
                       device=device).to(gnn.activation_dtype(precision))
  labels = torch.randint(0, num_classes, (batch,), generator=gen, device=device)
  return images, labels
