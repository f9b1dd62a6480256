"""Layer specifications of the benchmark workloads -- one source for both the
model builders and the mask tables (names / HWIO shapes / creation order) that
``get_mask_init_fn`` and the golden sparsity vectors are keyed by.

Topologies re-stated from the reference model files:
  ResNet-50 v1.5   rigl/imagenet_resnet/resnet_model.py:396-501, 577-731, 780-783
  MobileNet-v1     rigl/imagenet_resnet/mobilenetv1_model.py:156-342
  WideResNet       rigl/cifar_resnet/resnet_model.py:70-235
  MNIST MLP        rigl/mnist/mnist_train_eval.py:112-132
"""
from collections import OrderedDict, namedtuple

SCOPE = 'resnet_model'   # default variable scope of all three conv nets

ConvSpec = namedtuple('ConvSpec', 'end_point k cin cout stride role')
# role: 'stem' | 'proj' | 'c1' | 'c2' | 'c3'


def resnet50_blocks(width=1.0):
  """Yields (group, block_index, [ConvSpec...]) in variable-creation order."""
  blocks = [3, 4, 6, 3]
  in_ch = int(64 * width)
  for g in range(1, 5):
    f = int(64 * 2**(g - 1) * width)
    stride = 1 if g == 1 else 2
    name = 'block_group%d' % g
    ep = 'block_group_projection_%s' % name
    yield g, 0, [
        ConvSpec('bottleneck_projection_%s' % ep, 1, in_ch, 4 * f, stride, 'proj'),
        ConvSpec('bottleneck_1_%s' % ep, 1, in_ch, f, 1, 'c1'),
        ConvSpec('bottleneck_2_%s' % ep, 3, f, f, stride, 'c2'),
        ConvSpec('bottleneck_3_%s' % ep, 1, f, 4 * f, 1, 'c3'),
    ]
    for n in range(1, blocks[g - 1]):
      ep = '%s_%d_1' % (name, n)
      yield g, n, [
          ConvSpec('bottleneck_1_%s' % ep, 1, 4 * f, f, 1, 'c1'),
          ConvSpec('bottleneck_2_%s' % ep, 3, f, f, 1, 'c2'),
          ConvSpec('bottleneck_3_%s' % ep, 1, f, 4 * f, 1, 'c3'),
      ]
    in_ch = 4 * f


def resnet50_masks(prune_first_layer=True, prune_last_layer=True,
                   num_classes=1000, width=1.0):
  d = OrderedDict()
  fmt = SCOPE + '/%s/mask:0'
  if prune_first_layer:
    d[fmt % 'initial_conv'] = (7, 7, 3, int(64 * width))
  for _, _, convs in resnet50_blocks(width):
    for c in convs:
      d[fmt % c.end_point] = (c.k, c.k, c.cin, c.cout)
  if prune_last_layer:
    d[fmt % 'final_dense'] = (int(2048 * width), num_classes)
  return d


MOBILENET_V1_BLOCKS = [  # (filters, stride) for block_id 0..12
    (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1),
    (512, 1), (512, 1), (512, 1), (512, 1), (1024, 2), (1024, 1)]


def mobilenet_v1_masks(prune_last_layer=True, num_classes=1000):
  """Only the 13 pointwise convs (+ final_dense) are masked; the stem and the
  depthwise convs are dense."""
  d = OrderedDict()
  fmt = SCOPE + '/%s/mask:0'
  in_ch = 32
  for i, (f, _) in enumerate(MOBILENET_V1_BLOCKS):
    d[fmt % ('contraction_1x1_%d' % i)] = (1, 1, in_ch, f)
    in_ch = f
  if prune_last_layer:
    d[fmt % 'final_dense'] = (1024, num_classes)
  return d


def wide_resnet_convs(depth, width):
  """Yields (name, k, cin, cout, stride, kind) in creation order; depth must
  be 6n+4 (rigl/cifar_resnet/resnet_model.py:90-93)."""
  if (depth - 4) % 6 != 0:
    raise ValueError('Depth of ResNet specified not sufficient.')
  n_blocks = (depth - 4) // 6
  in_ch = 16
  for name, base, subsample in (('conv_2', 16, False), ('conv_3', 32, True),
                                ('conv_4', 64, True)):
    out = base * width
    for n in range(n_blocks):
      stride = 2 if (subsample and n == 0) else 1
      if in_ch != out:
        yield ('skip_%s' % name, 1, in_ch, out, stride, 'skip')
      yield ('%s_%d_1' % (name, n), 3, in_ch, out, stride, 'conv1')
      yield ('%s_%d_2' % (name, n), 3, out, out, 1, 'conv2')
      in_ch = out


def wide_resnet_masks(depth=22, width=1, prune_first_layer=False,
                      prune_last_layer=True, num_classes=10):
  d = OrderedDict()
  fmt = SCOPE + '/%s/mask:0'
  if prune_first_layer:
    d[fmt % 'conv_1'] = (3, 3, 3, 16)
  for name, k, cin, cout, _, _ in wide_resnet_convs(depth, width):
    d[fmt % name] = (k, k, cin, cout)
  if prune_last_layer:
    d[fmt % 'logits'] = (64 * width, num_classes)
  return d


def mnist_mlp_masks():
  return OrderedDict([('layer1/mask:0', (784, 300)), ('layer2/mask:0', (300, 100)),
                      ('layer3/mask:0', (100, 10))])


def resnet50_macs_per_image(num_classes=1000):
  """Dense MACs per image: (fwd, dgrad) -- dgrad skips the stem (no dX for the
  input images).  4 089 284 608 fwd (rigl/str_sparsities.py:29)."""
  stem = 7 * 7 * 3 * 64 * 112 * 112
  fwd = stem
  hw = 56
  for _, _, convs in resnet50_blocks():
    stride = [c.stride for c in convs if c.role == 'c2'][0]
    out = hw // stride
    for c in convs:
      res = hw if c.role == 'c1' else out
      fwd += c.k * c.k * c.cin * c.cout * res * res
    hw = out
  fwd += 2048 * num_classes
  return fwd, fwd - stem


def resnet50_layerwise_bound(batch, mfma_flops=2.5e15, hbm_bytes=8.0e12, num_classes=1000):
  """Per-layer roofline of the conv path, summed: every conv's fwd, dgrad and wgrad is charged
  max(2*MACs / MFMA peak, algorithmic bytes / HBM peak) -- bf16 activations in and out once,
  bf16 weights (fp32 dW for wgrad).  ResNet-50's 1x1 convs sit below the machine balance
  (312 flop/B), so the whole-net bound is well above total-flops / MFMA-peak.
  Returns (bound_s, mfma_only_s, hbm_only_s) for one step of `batch` images on one GPU."""
  layers = [(7, 3, 64, 224, 112, False)]          # (k, cin, cout, in_res, out_res, has_dgrad)
  hw = 56
  for _, _, convs in resnet50_blocks():
    stride = [c.stride for c in convs if c.role == 'c2'][0]
    out = hw // stride
    for c in convs:
      res_out = hw if c.role == 'c1' else out
      res_in = res_out if c.k == 1 else hw        # a strided 1x1 only needs the rows it samples
      layers.append((c.k, c.cin, c.cout, res_in, res_out, True))
    hw = out
  layers.append((1, 2048, num_classes, 1, 1, True))
  bound = t_mfma = t_hbm = 0.0
  for k, cin, cout, ri, ro, has_dgrad in layers:
    flops = 2.0 * batch * ro * ro * k * k * cin * cout
    act = 2.0 * batch * (ri * ri * cin + ro * ro * cout)
    w = k * k * cin * cout
    for wbytes, on in ((2 * w, True), (2 * w, has_dgrad), (4 * w, True)):   # fwd, dgrad, wgrad
      if not on:
        continue
      tc, tm = flops / mfma_flops, (act + wbytes) / hbm_bytes
      bound += max(tc, tm)
      t_mfma += tc
      t_hbm += tm
  return bound, t_mfma, t_hbm


def resnet50_stat_layers(num_classes=1000, width=1.0):
  """`sparse_utils.StatLayer`s of ResNet-50 in creation order (every conv + the classifier), for `sparse_utils.get_stats`."""
  from rigl_amd.sparse_utils import StatLayer  # pylint: disable=import-outside-toplevel
  out = [StatLayer('conv2d', 'initial_conv', (7, 7, 3, int(64 * width)), 224, (2, 2))]
  hw = 56
  for _, _, convs in resnet50_blocks(width):
    stride = [c.stride for c in convs if c.role == 'c2'][0]
    o = hw // stride
    for c in convs:
      res = o if c.role == 'c3' else hw
      out.append(StatLayer('conv2d', c.end_point, (c.k, c.k, c.cin, c.cout), res, (c.stride, c.stride)))
    hw = o
  out.append(StatLayer('dense', 'final_dense', (int(2048 * width), num_classes)))
  return out


def mobilenet_v1_stat_layers(num_classes=1000):
  """StatLayers of MobileNet-v1: stem, 13 x (depthwise 3x3, pointwise 1x1), classifier."""
  from rigl_amd.sparse_utils import StatLayer  # pylint: disable=import-outside-toplevel
  out = [StatLayer('conv2d', 'initial_conv', (3, 3, 3, 32), 224, (2, 2))]
  hw, in_ch = 112, 32
  for i, (f, s) in enumerate(MOBILENET_V1_BLOCKS):
    out.append(StatLayer('depthwise', 'depthwise_nxn_%d' % i, (3, 3, in_ch, 1), hw, (s, s)))
    hw //= s
    out.append(StatLayer('conv2d', 'contraction_1x1_%d' % i, (1, 1, in_ch, f), hw, (1, 1)))
    in_ch = f
  out.append(StatLayer('dense', 'final_dense', (1024, num_classes)))
  return out
