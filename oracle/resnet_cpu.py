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
  """NCHW; weights HWIO like the reference; masks float 0/1.  ``dtype``: torch.float32 (the reference's default
  --precision, and what the CPU baseline times) or torch.float64 (the exact-arithmetic stand-in the fp32 kernels are
  held against, tests/test_k1_fp32_gpu.py); the initial values are the float32 draws either way."""

  def __init__(self, sparsity_by_layer=None, seed=0, num_classes=1000, dtype=torch.float32):
    rs = np.random.RandomState(seed)
    self.w, self.m, self.bn = [], [], []
    self.dtype = dtype

    def conv(k, cin, cout):
      std = np.sqrt(1.0 / (k * k * cin))
      self.w.append(torch.from_numpy((rs.randn(k, k, cin, cout) * std).astype(np.float32)).to(dtype).requires_grad_(True))
      self.m.append(torch.ones(k, k, cin, cout, dtype=dtype))
      return len(self.w) - 1

    def bn(c, zero=False):
      g = torch.zeros(c, dtype=dtype) if zero else torch.ones(c, dtype=dtype)
      self.bn.append((g.requires_grad_(True), torch.zeros(c, dtype=dtype, requires_grad=True)))
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
    self.w[self.fc] = (torch.randn(1, 1, 2048, num_classes) * 0.01).to(dtype).requires_grad_(True)
    self.fc_b = torch.zeros(num_classes, dtype=dtype, requires_grad=True)
    if sparsity_by_layer is not None:
      for i, s in enumerate(sparsity_by_layer):
        self.m[i] = torch.from_numpy(O.get_mask_random_numpy(tuple(self.w[i].shape), s, rs).astype(np.float32)).to(dtype)
    self.mom = [torch.zeros_like(w) for w in self.w]
    self.mom_other = None      # momentum of the batch-norm parameters and the fc bias, created by the first step
    self._keep = None            # {layer: mask*W tensor} while a step is asked to keep the dense gradients
    self.dense_grads = None      # [dL/d(mask*W) as NumPy arrays] of the last such step

  def _conv(self, x, idx, stride):
    wm = self.m[idx] * self.w[idx]                             # y = conv(x, mask*W)
    if self._keep is not None:
      wm.retain_grad()                                         # dL/d(mask*W): RigL's dense gradient (a by-product)
      self._keep[idx] = wm
    w = wm.permute(3, 2, 0, 1)
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
    wfc = self.m[self.fc] * self.w[self.fc]
    if self._keep is not None:
      wfc.retain_grad()
      self._keep[self.fc] = wfc
    return x @ wfc.reshape(2048, -1) + self.fc_b

  def other_params(self):
    """Batch-norm gammas / betas and the fc bias: trained by the same MomentumOptimizer, no regulariser."""
    return [p for gb in self.bn for p in gb] + [self.fc_b]

  def train_step(self, images, labels, lr=0.1, mu=0.9, wd=1e-4, keep_dense=False):
    """fwd + bwd + masked Nesterov-momentum update (non-update iteration) of EVERY trainable variable
    (tf.train.MomentumOptimizer(use_nesterov=True), imagenet_train_eval.py:360-361; the l2 regulariser is attached to
    the conv / fc kernels only, resnet_model.py:287-295, 718-724)."""
    for w in self.w:
      w.grad = None
    for p in self.other_params():
      p.grad = None
    self._keep = {} if keep_dense else None
    loss = F.cross_entropy(self.forward(images), labels, label_smoothing=0.1)
    loss.backward()
    if keep_dense:
      self.dense_grads = [self._keep[i].grad.detach().numpy().copy() for i in range(len(self.w))]
      self._keep = None
    with torch.no_grad():
      for i, w in enumerate(self.w):
        # w.grad already = mask * dense (autograd through mask*W); + l2 on raw W
        g = w.grad + wd * w
        self.mom[i].mul_(mu).add_(g)
        w.sub_(lr * g + lr * mu * self.mom[i])
      others = self.other_params()
      if self.mom_other is None:
        self.mom_other = [torch.zeros_like(p) for p in others]
      for p, a in zip(others, self.mom_other):
        a.mul_(mu).add_(p.grad)
        p.sub_(lr * p.grad + lr * mu * a)
    return float(loss.detach())

  def mask_update(self, dense_grads, drop_fraction=0.3):
    """Full-sort prune/regrow on every layer (the reference's cost model:
    two O(n log n) sorts per layer)."""
    for i, w in enumerate(self.w):
      r = O.rigl_mask_update(self.m[i].numpy(), w.detach().numpy(), dense_grads[i], drop_fraction,
                             momentum=self.mom[i].numpy())
      self.m[i] = torch.from_numpy(r['mask']).to(self.dtype)
      with torch.no_grad():
        w.copy_(torch.from_numpy(r['weights']))
      self.mom[i] = torch.from_numpy(r['momentum'])


def time_cpu_baseline(batch=32, warmup=2, steps=5, threads=None, budget_s=90.0, sparsity=0.8):
  """The protocol of BASELINE.md section 3 / SURVEY 8(d): ResNet-50 at ``sparsity``, synthetic 224x224x3 N(0,1)
  images, batch 32, 2 warm-up + 5 timed steps (fwd + bwd + masked Nesterov update) and ONE real whole-model mask
  update (all 54 layers, full-sort prune/regrow fed the step's own dense gradients), timed separately and
  amortised over the update period of 100 steps.  Returns dict(value img/s, cores = threads used, host_cores, ...).
  Bounded: the timed steps stop early once ``budget_s`` is used up (at least one is always taken).
  Threads are capped at 32 -- torch-CPU convolutions of these batch sizes get SLOWER beyond that on the many-core
  hosts of the GPU boxes (256 threads: 200 s per batch-8 step)."""
  import os
  host_cores = os.cpu_count() or 1
  threads = threads or min(host_cores, 32)
  torch.set_num_threads(threads)
  model = ResNet50CPU(sparsity_by_layer=[sparsity] * 54)
  model.train_step(torch.randn(1, 3, 64, 64), torch.randint(0, 1000, (1,)))   # allocator, thread pool
  x = torch.randn(batch, 3, 224, 224)
  y = torch.randint(0, 1000, (batch,))
  t_all = time.time()
  n_warm = 0
  for _ in range(warmup):
    model.train_step(x, y)
    n_warm += 1
    if time.time() - t_all > budget_s * 0.25:
      break
  t0 = time.time()
  n = 0
  while n < steps and (n == 0 or time.time() - t_all < budget_s * 0.6):
    model.train_step(x, y, keep_dense=(n == steps - 1))
    n += 1
  t_step = (time.time() - t0) / n
  # one real mask update of the whole model: |mask*W| prune + |dense grad| regrow on every layer
  if model.dense_grads is None:
    model.train_step(x, y, keep_dense=True)      # (the budget cut the loop short of its last, gradient-keeping step)
  t1 = time.time()
  model.mask_update(model.dense_grads, 0.3)
  t_update = time.time() - t1
  ips = batch / (t_step + t_update / 100.0)
  return dict(value=ips, unit='images/sec', cores=threads, host_cores=host_cores, kind='port',
              s_per_step=t_step, s_per_mask_update=t_update, batch=batch, timed_steps=n, warmup_steps=n_warm,
              sample='ResNet-50 (uniform %.2f masks) fp32 torch-CPU dense conv2d(x, mask*W) fwd+bwd+Nesterov, batch %d: %d warm-up + '
                     '%d timed steps on %d threads of a %d-core host; one real full-sort mask update of all 54 layers '
                     '(NumPy restatement of sparse_optimizers_base.py:276-343) timed separately and amortised over 100 steps; '
                     'TensorFlow itself is not installable here (tests/tf_reference/run_tf_cpu.py is the script for a TF-1.15 box)'
                     % (sparsity, batch, n_warm, n, threads, host_cores))
