"""MNIST MLP 784-300-100-10 from masked fully-connected layers --
rigl/mnist/mnist_train_eval.py:112-132 (BASELINE config 1: 90 % sparsity on
layers 1-2, layer 3 forced dense by the script, :269-272).  ReLU on the first
two layers, biases unmasked, l2 1e-4.  300/100/10 units are not multiples of 8,
so these layers run on the direct kernels (conv_ref.hip)."""
import torch

from rigl_amd import pruning_layers as PL
from rigl_amd import variables as V
from rigl_amd.workloads import nn as gnn


class MnistMLP:

  def __init__(self, graph=None, pruning_method='threshold', l2_scale=1e-4, seed=0):
    self.graph = g = graph or V.get_default_graph()
    PL.set_init_seed(seed)
    self.layers = []
    n_in = 784
    for i, units in enumerate((300, 100, 10), start=1):
      scope = 'layer%d' % i
      layer = PL.MaskedDense(g, scope, n_in, units, True, pruning_method, l2_scale, None,
                             torch.relu if i < 3 else None)
      g.modules[scope] = layer
      self.layers.append(layer)
      n_in = units
    g.finalize()

  def __call__(self, x):
    for l in self.layers:
      x = l(x)
    return x

  def loss(self, images, labels):
    # tf.losses.sparse_softmax_cross_entropy (mnist_train_eval.py:151-154)
    return gnn.softmax_cross_entropy(self(images), labels, 0.0)


def synthetic_batch(batch, device, seed=1234):
  gen = torch.Generator(device=device).manual_seed(seed)
  images = torch.rand(batch, 784, generator=gen, device=device).to(torch.bfloat16)
  labels = torch.randint(0, 10, (batch,), generator=gen, device=device)
  return images, labels
