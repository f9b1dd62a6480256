"""CPU restatement of the reference's ResNet-50 RigL training step in fp32
PyTorch-CPU ops: dense storage, dense compute -- ``conv2d(x, mask * W)`` exactly
as TF's masked_conv2d computes it (rigl/imagenet_resnet/pruning_layers.py
:139-157), Nesterov momentum with l2 on the raw weights, and the full-sort
(k = n) prune/regrow update of rigl/sparse_optimizers_base.py:276-343 via the
NumPy oracle.

TEST INFRASTRUCTURE / CPU BASELINE ONLY ("port": TensorFlow itself cannot be
installed here -- no network).  Used by bench.py's ``cpu_baseline`` leg and by
the end-to-end parity tests; never by the product.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from oracle import rigl_oracle as O


def resnet50_blocks():
  blocks = [3, 4, 6, 3]
  in_ch = 64
  for g in range(1, 5):
    f = 64 * 2**(g - 1)
    for n in range(blocks[g - 1]):
      stride = 2 if (g > 1 and n == 0) else 1
      yield in_ch, f, stride, n == 0
      in_ch = 4 * f


class ResNet50CPU:
  """fp32 NCHW; weights HWIO like the reference; masks float 0/1."""

  def __init__(self, sparsity_by_layer=None, seed=0, num_classes=1000):
    rs = np.random.RandomState(seed)
    self.w, self.m, self.bn = [], [], []

    def conv(k, cin, cout):
      std = np.sqrt(1.0 / (k * k * cin))
      self.w.append(torch.from_numpy((rs.randn(k, k, cin, cout) * std).astype(np.float32)).requires_grad_(True))
      self.m.append(torch.ones(k, k, cin, cout))
      return len(self.w) - 1

    def bn(c, zero=False):
      g = torch.zeros(c) if zero else torch.ones(c)
      self.bn.append((g.requires_grad_(True), torch.zeros(c, requires_grad=True)))
      return len(self.bn) - 1

    self.stem = (conv(7, 3, 64), bn(64))
    self.blocks = []
    for cin, f, stride, proj in resnet50_blocks():
      b = {}
      if proj:
        b['proj'] = (conv(1, cin, 4 * f), bn(4 * f))
      b['c1'] = (conv(1, cin, f), bn(f))
      b['c2'] = (conv(3, f, f), bn(f))
      b['c3'] = (conv(1, f, 4 * f), bn(4 * f, zero=True))
      b['stride'] = stride
      self.blocks.append(b)
    self.fc = conv(1, 2048, num_classes)
    self.w[self.fc] = (torch.randn(1, 1, 2048, num_classes) * 0.01).requires_grad_(True)
    self.fc_b = torch.zeros(num_classes, requires_grad=True)
    if sparsity_by_layer is not None:
      for i, s in enumerate(sparsity_by_layer):
        self.m[i] = torch.from_numpy(O.get_mask_random_numpy(tuple(self.w[i].shape), s, rs).astype(np.float32))
    self.mom = [torch.zeros_like(w) for w in self.w]
    self.mom_other = None

  def _conv(self, x, idx, stride):
    w = (self.m[idx] * self.w[idx]).permute(3, 2, 0, 1)       # y = conv(x, mask*W)
    k = w.shape[-1]
    if stride > 1:
      pad = k - 1
      x = F.pad(x, (pad // 2, pad - pad // 2, pad // 2, pad - pad // 2))
      return F.conv2d(x, w, stride=stride)
    return F.conv2d(x, w, padding=k // 2)

  def _bn(self, x, idx, relu):
    g, b = self.bn[idx]
    y = F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)
    return F.relu(y) if relu else y

  def forward(self, x):
    x = self._bn(self._conv(x, self.stem[0], 2), self.stem[1], True)
    x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float('-inf')), 3, 2)
    for b in self.blocks:
      sc = x
      if 'proj' in b:
        sc = self._bn(self._conv(x, b['proj'][0], b['stride']), b['proj'][1], False)
      y = self._bn(self._conv(x, b['c1'][0], 1), b['c1'][1], True)
      y = self._bn(self._conv(y, b['c2'][0], b['stride']), b['c2'][1], True)
      y = self._bn(self._conv(y, b['c3'][0], 1), b['c3'][1], False)
      x = F.relu(y + sc)
    x = x.mean(dim=(2, 3))
    wfc = (self.m[self.fc] * self.w[self.fc]).reshape(2048, -1)
    return x @ wfc + self.fc_b

  def train_step(self, images, labels, lr=0.1, mu=0.9, wd=1e-4):
    """fwd + bwd + masked Nesterov-momentum update (non-update iteration)."""
    for w in self.w:
      w.grad = None
    loss = F.cross_entropy(self.forward(images), labels, label_smoothing=0.1)
    loss.backward()
    with torch.no_grad():
      for i, w in enumerate(self.w):
        # w.grad already = mask * dense (autograd through mask*W); + l2 on raw W
        g = w.grad + wd * w
        self.mom[i].mul_(mu).add_(g)
        w.sub_(lr * g + lr * mu * self.mom[i])
      for g_, b_ in self.bn:
        for p in (g_, b_):
          p.sub_(lr * p.grad)
          p.grad = None
    return float(loss)

  def mask_update(self, dense_grads, drop_fraction=0.3):
    """Full-sort prune/regrow on every layer (the reference's cost model:
    two O(n log n) sorts per layer)."""
    for i, w in enumerate(self.w):
      r = O.rigl_mask_update(self.m[i].numpy(), w.detach().numpy(), dense_grads[i], drop_fraction,
                             momentum=self.mom[i].numpy())
      self.m[i] = torch.from_numpy(r['mask'])
      with torch.no_grad():
        w.copy_(torch.from_numpy(r['weights']))
      self.mom[i] = torch.from_numpy(r['momentum'])


def time_cpu_baseline(batch=8, steps=2, threads=None, budget_s=25.0):
  """Returns dict(value img/s incl. amortised mask update, cores, sample).
  Bounded: one tiny warm-up step, then at most `steps` timed steps or
  `budget_s` seconds.  Threads are capped at 32 -- torch-CPU convolutions of a
  batch-8 problem get SLOWER beyond that (256 threads: 200 s per step)."""
  import os
  threads = threads or min(os.cpu_count() or 1, 32)
  torch.set_num_threads(threads)
  model = ResNet50CPU(sparsity_by_layer=None)
  model.train_step(torch.randn(1, 3, 64, 64), torch.randint(0, 1000, (1,)))   # warm-up (allocator, thread pool)
  x = torch.randn(batch, 3, 224, 224)
  y = torch.randint(0, 1000, (batch,))
  t0 = time.time()
  n = 0
  while n < steps and (n == 0 or time.time() - t0 < budget_s * 0.6):
    model.train_step(x, y)
    n += 1
  t_step = (time.time() - t0) / n
  # mask update: time the two largest + a few small layers, scale by element count
  sizes = [int(w.numel()) for w in model.w]
  probe = sorted(range(len(sizes)), key=lambda i: -sizes[i])[:2] + [1, 2, 3]
  t1 = time.time()
  done = 0
  for i in probe:
    w = model.w[i].detach().numpy()
    g = np.random.randn(*w.shape).astype(np.float32)
    O.rigl_mask_update(model.m[i].numpy(), w, g, 0.3)
    done += sizes[i]
  t_update = (time.time() - t1) * (sum(sizes) / done)
  ips = batch / (t_step + t_update / 100.0)
  return dict(value=ips, unit='images/sec', cores=threads, kind='port',
              s_per_step=t_step, s_per_mask_update=t_update,
              sample='ResNet-50 fp32 torch-CPU dense conv2d(x, mask*W) fwd+bwd+Nesterov, batch %d x %d '
                     'steps on %d threads; full-sort mask update timed on %d of 54 layers and scaled by '
                     'weight count, amortised over 100 steps' % (batch, n, threads, len(probe)))
