"""Mask names and HWIO shapes of the four benchmark workloads, written down
from the reference model builders (test-side table; the product builds the
same lists from its own workload definitions and the tests compare them).

  ResNet-50   rigl/imagenet_resnet/resnet_model.py:396-501,577-731,780-783
  MobileNet   rigl/imagenet_resnet/mobilenetv1_model.py:156-342 (masked = 13
              pointwise convs + final_dense; depthwise and stem are dense)
  WRN-d-w     rigl/cifar_resnet/resnet_model.py:70-235
  MNIST MLP   rigl/mnist/mnist_train_eval.py:112-132
"""
from collections import OrderedDict


def resnet50(prune_first_layer=True, prune_last_layer=True, num_classes=1000):
  d = OrderedDict()
  scope = 'resnet_model/%s/mask:0'
  if prune_first_layer:
    d[scope % 'initial_conv'] = (7, 7, 3, 64)
  blocks = [3, 4, 6, 3]
  in_ch = 64
  for g in range(1, 5):
    f = 64 * 2**(g - 1)
    name = 'block_group%d' % g
    ep = 'block_group_projection_%s' % name
    d[scope % ('bottleneck_projection_%s' % ep)] = (1, 1, in_ch, 4 * f)
    d[scope % ('bottleneck_1_%s' % ep)] = (1, 1, in_ch, f)
    d[scope % ('bottleneck_2_%s' % ep)] = (3, 3, f, f)
    d[scope % ('bottleneck_3_%s' % ep)] = (1, 1, f, 4 * f)
    for n in range(1, blocks[g - 1]):
      ep = '%s_%d_1' % (name, n)
      d[scope % ('bottleneck_1_%s' % ep)] = (1, 1, 4 * f, f)
      d[scope % ('bottleneck_2_%s' % ep)] = (3, 3, f, f)
      d[scope % ('bottleneck_3_%s' % ep)] = (1, 1, f, 4 * f)
    in_ch = 4 * f
  if prune_last_layer:
    d[scope % 'final_dense'] = (2048, num_classes)
  return d


def mobilenet_v1(prune_last_layer=True, num_classes=1000):
  d = OrderedDict()
  scope = 'resnet_model/%s/mask:0'
  filters = [64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024]
  in_ch = 32
  for i, f in enumerate(filters):
    d[scope % ('contraction_1x1_%d' % i)] = (1, 1, in_ch, f)
    in_ch = f
  if prune_last_layer:
    d[scope % 'final_dense'] = (1024, num_classes)
  return d


def wide_resnet(depth=22, width=1, prune_first_layer=False,
                prune_last_layer=True, num_classes=10):
  assert (depth - 4) % 6 == 0
  n_blocks = (depth - 4) // 6
  d = OrderedDict()
  scope = 'resnet_model/%s/mask:0'
  if prune_first_layer:
    d[scope % 'conv_1'] = (3, 3, 3, 16)
  in_ch = 16
  for name, base in (('conv_2', 16), ('conv_3', 32), ('conv_4', 64)):
    out = base * width
    for n in range(n_blocks):
      if in_ch != out:
        d[scope % ('skip_%s' % name)] = (1, 1, in_ch, out)
      d[scope % ('%s_%d_1' % (name, n))] = (3, 3, in_ch, out)
      d[scope % ('%s_%d_2' % (name, n))] = (3, 3, out, out)
      in_ch = out
  if prune_last_layer:
    d[scope % 'logits'] = (64 * width, num_classes)
  return d


def mnist_mlp():
  d = OrderedDict()
  d['layer1/mask:0'] = (784, 300)
  d['layer2/mask:0'] = (300, 100)
  d['layer3/mask:0'] = (100, 10)
  return d
