"""Dynamic-sparse-training optimizers with the reference's API
(rigl/sparse_optimizers.py + rigl/sparse_optimizers_base.py), executing the
mask update on the fused HIP prune/regrow kernels (K2).

Same constructor arguments, attribute names (``drop_fraction``,
``_weight2masked_grads``), method names and ValueErrors as the reference, so
the parity tests read like ``rigl/sparse_optimizers_test.py``.  It is an EAGER
twin: one ``minimize`` / ``apply_gradients`` call is one training step.

Step semantics:
  * RigL  (sparse_optimizers_base.py:487-521): on a mask-update iteration the
    masks change and nothing else happens -- no gradient apply, global_step not
    incremented; otherwise the inner optimizer applies and increments.
  * SET / Static / Momentum (:118-146): the inner optimizer is applied every
    step, then the mask is maybe updated with the incremented step.
The schedule is scalar host logic on python ints / float32; everything that
touches a weight runs in the HIP kernels.
"""
import math
import re

import numpy as np
import torch

from rigl_amd import _lib
from rigl_amd import ops
from rigl_amd import pyhash
from rigl_amd import train
from rigl_amd import variables as V

F32 = np.float32


def extract_number(token):
  """Trailing ``_<number>`` of a method string, else 1 (base.py:45-59)."""
  found = re.search(r'.*_(\d*\.?\d*)$', token)
  return float(found.group(1)) if found else 1.


def _stable_hash(text):
  """The reference's ``hash(weights.name + 'drop')`` seed: the
  builtin hash under a fixed PYTHONHASHSEED, else its PYTHONHASHSEED=0 value."""
  return pyhash.name_hash(text)


class PruningGetterMixin:
  """PruningGetterTf1Mixin (sparse_optimizers.py:46-56): the masked layers of
  the graph in creation order."""

  def get_weights(self):
    return self.graph.get_weights()

  def get_masks(self):
    return self.graph.get_masks()

  def get_masked_weights(self):
    return self.graph.get_weights()   # d/d(mask*W): the dense gradient slot


class SparseSETOptimizerBase(train.Optimizer):
  """SET (base.py:62-419): drop by magnitude, grow at random."""

  def __init__(self, optimizer, begin_step, end_step, frequency,
               drop_fraction=0.1, drop_fraction_anneal='constant',
               use_locking=False, grow_init='zeros', name='SparseSETOptimizer',
               use_stateless=True, stateless_seed_offset=0, noise_std=1e-5):
    super().__init__(use_locking, name, getattr(optimizer, '_graph', None))
    self._optimizer = optimizer
    self._grow_init = grow_init
    self._drop_fraction_anneal = drop_fraction_anneal
    self._drop_fraction_initial_value = F32(float(drop_fraction))
    self._begin_step = int(begin_step)
    self._end_step = int(end_step)
    self._frequency = int(frequency)
    self._frequency_val = int(frequency)
    self._use_stateless = use_stateless
    self._stateless_seed_offset = int(stateless_seed_offset)
    self._noise_std = float(noise_std)
    self._last_update_step = -self._frequency_val          # :166-171
    self._global_step = None
    self.drop_fraction = F32(0.)
    self.last_counts = None
    self._reinit_when_same = False
    self._momentum_reset_mode = _lib.MOMRESET_ZEROS          # :345-353

  # ---- tf.train.Optimizer surface --------------------------------------------
  def compute_gradients(self, loss, **kwargs):
    return self._optimizer.compute_gradients(loss, **kwargs)

  def get_slot_names(self):
    return self._optimizer.get_slot_names()

  def get_slot(self, var, name):
    return self._optimizer.get_slot(var, name)

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    """:118-146."""
    self._before_apply_gradients(grads_and_vars)
    self._optimizer.apply_gradients(grads_and_vars, global_step=global_step,
                                    name=name)
    global_step = (global_step if global_step is not None else
                   train.get_or_create_global_step(self.graph))
    self._global_step = global_step
    return self.cond_mask_update_op(global_step, lambda: None)

  def _before_apply_gradients(self, grads_and_vars):
    del grads_and_vars

  def cond_mask_update_op(self, global_step, false_branch):
    """:152-187."""
    gs = int(global_step.value)
    if self.is_mask_update_iter(gs, self._last_update_step):
      self.mask_update_op(gs)
      self._last_update_step = gs
      return True
    false_branch()
    return False

  def mask_update_op(self, gs):
    """All layers in ONE batched K2 call (the reference loops layers,
    :173-176)."""
    del gs
    mls = self.graph.masked_layers()
    self._prefetch_drop_noise(mls)          # every layer's noise tensor in one launch
    try:
      layers = [self._layer_request(l) for l in mls]
    finally:
      self._noise_cache = {}
    self._run_update(layers)

  # ---- schedule (scalar host logic) ------------------------------------------
  def is_mask_update_iter(self, global_step, last_update_step):
    """:198-230; also sets ``self.drop_fraction``."""
    gs = int(global_step.value) if hasattr(global_step, 'value') else int(global_step)
    in_range = gs >= self._begin_step and (gs <= self._end_step or
                                           self._end_step < 0)
    is_update = in_range and (int(last_update_step) + self._frequency <= gs)
    self.drop_fraction = self.get_drop_fraction(gs, is_update)
    return bool(is_update)

  def get_drop_fraction(self, global_step, is_mask_update_iter_op):
    """:232-258, float32 like the TF graph."""
    gs = int(global_step.value) if hasattr(global_step, 'value') else int(global_step)
    init = self._drop_fraction_initial_value
    anneal = self._drop_fraction_anneal
    if anneal == 'constant':
      frac = init
    elif anneal == 'cosine':
      decay_steps = F32(self._end_step - self._begin_step)
      step = np.minimum(F32(gs), decay_steps)
      completed = F32(step / decay_steps)
      cosine = F32(F32(0.5) * F32(F32(1.0) + F32(np.cos(F32(F32(math.pi) * completed)))))
      frac = F32(init * cosine)
    elif isinstance(anneal, str) and anneal.startswith('exponential'):
      exponent = extract_number(anneal)
      power = F32(F32(gs - self._begin_step) / F32(self._end_step - self._begin_step))
      frac = F32(init * F32(np.power(F32(F32(1) - power), F32(exponent))))
    else:
      raise ValueError('drop_fraction_anneal: %s is not valid' % anneal)
    return F32(frac) if is_mask_update_iter_op else F32(0.)

  # ---- one-layer API of the reference ------------------------------------------
  def generic_mask_update(self, mask, weights, noise_std=1e-5):
    """:260-274 -- SET: |mask*W| + noise vs uniform-random grow scores."""
    lv = self._find_layer(mask, weights)
    self._run_update([self._layer_request(lv, noise_std=noise_std)])
    return mask

  def _get_update_op(self, score_drop, score_grow, mask, weights,
                     reinit_when_same=False):
    """:276-343 with explicit score tensors (same shape as the weights)."""
    lv = self._find_layer(mask, weights)
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits,
               momentum=self._slot_of(lv),
               score_drop=_flat_f32(score_drop, self.graph.device),
               score_grow=_flat_f32(score_grow, self.graph.device))
    self._attach_grow_values(req, lv)
    self._run_update([req], reinit_when_same=reinit_when_same)
    return mask

  def reset_momentum(self, weights, new_connections):
    """:345-353 stand-alone form (the fused kernel does this in-line)."""
    for s_name in self._optimizer.get_slot_names():
      slot = self._optimizer.get_slot(weights, s_name)
      slot[torch.as_tensor(new_connections, device=slot.device).bool()] = 0

  def get_grow_tensor(self, weights, method):
    """:355-400.  Returns a tensor shaped like ``weights``."""
    if not isinstance(method, str):
      raise ValueError('Grow-Init: %s is not a string' % method)
    w = weights.data if hasattr(weights, 'data') and not torch.is_tensor(weights) else weights
    if method == 'zeros':
      return torch.zeros_like(w)
    if method.startswith('initial_dist'):
      init = getattr(weights, 'initial_value', None)
      if init is None:
        raise ValueError('Grow-Init: initial_dist needs weights.initial_value')
      divisor = extract_number(method)
      # The reference shuffles with tf.random_shuffle (:372-380), a STATEFUL op whose stream hangs off the graph
      # seed and the op's position in the graph -- there is no stream to be bit-compatible with, only the
      # distribution: a uniform random permutation of the initial values.  Here the permutation is the stable
      # argsort of this layer's stateless uniform stream (seed = hash(name + 'grow_init_i'), global_step), i.e.
      # the same counter-based generator as every other random tensor of the update: identical on every replica
      # by construction and reproducible from (seed offset, step) alone.  Parity: distribution only.
      keys = self._random_uniform(tuple([init.numel()]), seed=self._seed(weights, 'grow_init_i'))
      perm = torch.argsort(keys.reshape(-1), stable=True)
      return init.reshape(-1)[perm].reshape(init.shape) / divisor
    if method.startswith('random_normal'):
      divisor = extract_number(method)
      std = w.float().std(unbiased=False)
      return self._random_normal(w.shape, std, w.dtype,
                                 self._seed(weights, 'grow_init_n')) / divisor
    if method.startswith('random_uniform'):
      divisor = extract_number(method)
      mean = w.abs().mean()
      return self._random_uniform(w.shape, -mean, mean, w.dtype,
                                  self._seed(weights, 'grow_init_u')) / divisor
    raise ValueError('Grow-Init: %s is not a valid option.' % method)

  # ---- randomness: stateless, keyed on (offset + name hash, global_step) ---------
  def _seed(self, weights, tag):
    name = getattr(weights, 'name', 'anonymous')
    return self._stateless_seed_offset + _stable_hash(name + tag)

  def _generator(self, weights, tag):
    gen = torch.Generator(device=self.graph.device)
    gs = int(self._global_step.value) if self._global_step is not None else 0
    gen.manual_seed((self._seed(weights, tag) * 1000003 + gs) & 0x7FFFFFFFFFFFFFFF)
    return gen

  def _tf_seed(self, seed):
    """int32([offset + seed, global_step]) -- the reference adds the offset in
    _random_normal/_random_uniform (:402-418); ``seed`` already carries it here."""
    gs = int(self._global_step.value) if self._global_step is not None else 0
    return int(seed), gs

  def _random_normal(self, shape, stddev, dtype, seed):
    """stateless_random_normal(shape, stddev=stddev, seed=[seed, global_step]) with
    TensorFlow's bit layout (Philox-4x32-10 + Box-Muller, rigl_stateless_random)."""
    n = 1
    for d in shape:
      n *= int(d)
    s0, s1 = self._tf_seed(seed)
    out = ops.stateless_random(n, s0, s1, 'normal', scale=float(stddev), shift=0.0,
                               device=self.graph.device)
    return out.view(tuple(shape)).to(dtype)

  def _random_uniform(self, shape, minval=0., maxval=1., dtype=torch.float32,
                      seed=0):
    n = 1
    for d in shape:
      n *= int(d)
    s0, s1 = self._tf_seed(seed)
    lo, hi = float(minval), float(maxval)
    # rnd * (maxval - minval) + minval with the difference taken in fp32, like the TF graph
    out = ops.stateless_random(n, s0, s1, 'uniform', scale=float(np.float32(hi) - np.float32(lo)), shift=lo,
                               device=self.graph.device)
    return out.view(tuple(shape)).to(dtype)

  # ---- internals ------------------------------------------------------------------
  def _find_layer(self, mask, weights):
    for l in self.graph.masked_layers():
      if l.weights is weights or l.mask is mask:
        return l
    raise ValueError('no masked layer owns %r / %r' % (mask, weights))

  def _slot_of(self, lv):
    names = self._optimizer.get_slot_names()
    if not names:
      return None
    return self._optimizer.get_slot(lv.weights, names[0]).view(-1)

  def _prefetch_drop_noise(self, layers):
    """stateless_random_normal(shape, stddev, seed=[offset + hash(name + 'drop'), global_step]) of every layer
    (:526-534) through rigl_stateless_random_batched: the same streams as the per-layer call, one launch."""
    self._noise_cache = {}
    std = self._noise_std
    if not std or not layers:
      return
    items = []
    for lv in layers:
      s0, s1 = self._tf_seed(self._seed(lv.weights, 'drop'))
      items.append((lv.weights.numel, s0, s1, 'normal', float(std), 0.0))
    outs = ops.stateless_random_batched(items, self.graph.device)
    self._noise_cache = {lv.weights.name: (float(std), t) for lv, t in zip(layers, outs)}

  def _drop_noise(self, lv, noise_std):
    if not noise_std:
      return None
    hit = getattr(self, '_noise_cache', {}).get(lv.weights.name)
    if hit is not None and hit[0] == float(noise_std):
      return hit[1]
    return self._random_normal(lv.weights.shape, noise_std, torch.float32,
                               self._seed(lv.weights, 'drop')).view(-1)

  def _layer_request(self, lv, noise_std=None):
    """SET: magnitude drop (+noise), uniform-random grow scores (:260-274)."""
    noise_std = self._noise_std if noise_std is None else noise_std
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits,
               momentum=self._slot_of(lv),
               drop_noise=self._drop_noise(lv, noise_std),
               score_grow=self._random_uniform(
                   lv.weights.shape, seed=self._seed(lv.weights, 'grow')).view(-1))
    self._attach_grow_values(req, lv)
    return req

  def _grow_mode(self):
    """(mode, divisor) of the fused kernel for self._grow_init."""
    method = self._grow_init
    if not isinstance(method, str):
      raise ValueError('Grow-Init: %s is not a string' % method)
    if method == 'zeros':
      return _lib.GROW_ZEROS, 1.0
    if method.startswith(('initial_dist', 'random_normal', 'random_uniform')):
      return _lib.GROW_EXPLICIT, 1.0
    raise ValueError('Grow-Init: %s is not a valid option.' % method)

  def _attach_grow_values(self, req, lv):
    mode, _ = self._grow_mode()
    if mode == _lib.GROW_EXPLICIT:
      req['grow_values'] = self.get_grow_tensor(
          lv.weights, self._grow_init).float().contiguous().view(-1)

  def _run_update(self, layers, reinit_when_same=None):
    mode, div = self._grow_mode()
    if reinit_when_same is None:
      reinit_when_same = self._reinit_when_same
    sync = getattr(self._optimizer, '_grad_sync', None)
    if sync is not None and getattr(sync, 'enabled', False):
      # data parallel: rank 0's value is authoritative (a 4-byte broadcast per mask update; the
      # fraction is a pure function of global_step, so this only guards against drifted hosts)
      from rigl_amd import dist as rdist  # pylint: disable=import-outside-toplevel
      self.drop_fraction = F32(rdist.broadcast_drop_fraction(float(self.drop_fraction), group=sync.group))
    self.last_counts = ops.prune_regrow(
        layers, float(self.drop_fraction), grow_init_mode=mode,
        grow_init_div=div, momentum_reset_mode=self._momentum_reset_mode,
        initial_acc_scale=getattr(self, '_initial_acc_scale', 0.0),
        reinit_when_same=reinit_when_same)
    self.graph.shadows_dirty = True
    if sync is not None and getattr(sync, 'enabled', False):
      from rigl_amd import dist as rdist  # pylint: disable=import-outside-toplevel
      if rdist.DEBUG and not sync.check_masks_identical():     # RIGL_DEBUG=1: replicas must hold ONE bitmap after every update
        raise RuntimeError('mask update left different masks on the replicas (RIGL_DEBUG check, '
                           'sparse_optimizers_base.py:471-476 expects identical summed gradients)')


def _flat_f32(t, device):
  if not torch.is_tensor(t):
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(t, dtype=np.float32)))
  return t.to(device, torch.float32).contiguous().view(-1)


class SparseRigLOptimizerBase(SparseSETOptimizerBase):
  """RigL (base.py:421-564): grow where the dense gradient is largest."""

  def __init__(self, optimizer, begin_step, end_step, frequency,
               drop_fraction=0.1, drop_fraction_anneal='constant',
               use_locking=False, grow_init='zeros', initial_acc_scale=0.,
               use_tpu=False, name='SparseRigLOptimizer',
               stateless_seed_offset=0, noise_std=1e-5):
    super().__init__(optimizer, begin_step, end_step, frequency,
                     drop_fraction=drop_fraction,
                     drop_fraction_anneal=drop_fraction_anneal,
                     grow_init=grow_init, use_locking=use_locking,
                     name='SparseRigLOptimizer',
                     stateless_seed_offset=stateless_seed_offset,
                     noise_std=noise_std)
    del name
    self._initial_acc_scale = float(initial_acc_scale)
    self._use_tpu = use_tpu     # replicas: the dense grads are all-reduced by
    #                             the inner optimizer's GradSync (rigl_amd.dist)
    self._momentum_reset_mode = _lib.MOMRESET_GRAD           # :555-564
    self._masked_grads = None
    self._weight2masked_grads = {}

  def set_masked_grads(self, grads, weights):
    """:471-476 (the cross-replica sum already happened in the arena)."""
    self._masked_grads = grads
    self._weight2masked_grads = {w.name: m for w, m in zip(weights, grads)}

  def compute_gradients(self, loss, **kwargs):
    """:478-485."""
    grads_and_vars = self._optimizer.compute_gradients(loss, **kwargs)
    masked_grads_vars = self._optimizer.compute_gradients(
        loss, var_list=self.get_masked_weights())
    self.set_masked_grads([g for g, _ in masked_grads_vars],
                          self.get_weights())
    return grads_and_vars

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    """:487-521."""
    self._before_apply_gradients(grads_and_vars)
    gs = (global_step if global_step is not None else
          train.get_or_create_global_step(self.graph))
    self._global_step = gs
    self._optimizer.graph.finalize()
    if hasattr(self._optimizer, '_ensure_slots'):
      self._optimizer._ensure_slots()   # "call this to create slots" (:505-506)

    def apply_gradient_op():
      self._optimizer.apply_gradients(grads_and_vars, global_step=global_step,
                                      name=name)

    return self.cond_mask_update_op(gs, apply_gradient_op)

  def generic_mask_update(self, mask, weights, noise_std=1e-5):
    """:523-538."""
    lv = self._find_layer(mask, weights)
    self._run_update([self._layer_request(lv, noise_std=noise_std)])
    return mask

  def _layer_request(self, lv, noise_std=None):
    noise_std = self._noise_std if noise_std is None else noise_std
    grad = self._weight2masked_grads.get(lv.weights.name)
    if grad is None:
      grad = lv.weights.grad
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits,
               momentum=self._slot_of(lv), dense_grad=grad.view(-1),
               drop_noise=self._drop_noise(lv, noise_std))
    self._attach_grow_values(req, lv)
    return req

  def _grow_mode(self):
    """:540-553."""
    method = self._grow_init
    if isinstance(method, str) and method.startswith('grad_scale'):
      return _lib.GROW_GRAD_SCALE, extract_number(method)
    if isinstance(method, str) and method.startswith('grad_sign'):
      return _lib.GROW_GRAD_SIGN, extract_number(method)
    return super()._grow_mode()

  def get_grow_tensor(self, weights, method):
    """:540-553."""
    if isinstance(method, str) and method.startswith('grad_scale'):
      return self._weight2masked_grads[weights.name] / extract_number(method)
    if isinstance(method, str) and method.startswith('grad_sign'):
      return torch.sign(self._weight2masked_grads[weights.name]) / extract_number(method)
    return super().get_grow_tensor(weights, method)

  def reset_momentum(self, weights, new_connections):
    """:555-564 stand-alone form."""
    nc = torch.as_tensor(new_connections, device=self.graph.device).bool()
    acc = self._weight2masked_grads[weights.name] * self._initial_acc_scale
    for s_name in self._optimizer.get_slot_names():
      slot = self._optimizer.get_slot(weights, s_name)
      slot[nc] = acc[nc]


class SparseSETOptimizer(PruningGetterMixin, SparseSETOptimizerBase):
  """sparse_optimizers.py:59-61."""


class SparseRigLOptimizer(PruningGetterMixin, SparseRigLOptimizerBase):
  """sparse_optimizers.py:64-66."""


class SparseStaticOptimizer(SparseSETOptimizer):
  """sparse_optimizers.py:69-123: score_grow = mask, reinit_when_same=True --
  the mask never changes, the weakest connections are re-initialised."""

  def __init__(self, optimizer, begin_step, end_step, frequency,
               drop_fraction=0.1, drop_fraction_anneal='constant',
               use_locking=False, grow_init='zeros',
               name='SparseStaticOptimizer', stateless_seed_offset=0,
               noise_std=1e-5):
    super().__init__(optimizer, begin_step, end_step, frequency,
                     drop_fraction=drop_fraction,
                     drop_fraction_anneal=drop_fraction_anneal,
                     grow_init=grow_init, use_locking=use_locking, name=name,
                     stateless_seed_offset=stateless_seed_offset,
                     noise_std=noise_std)
    self._reinit_when_same = True

  def _layer_request(self, lv, noise_std=None):
    noise_std = self._noise_std if noise_std is None else noise_std
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits,
               momentum=self._slot_of(lv),
               drop_noise=self._drop_noise(lv, noise_std),
               score_grow=lv.mask.data.view(-1))            # :121
    self._attach_grow_values(req, lv)
    return req


class SparseMomentumOptimizer(SparseSETOptimizer):
  """SNFS-style (sparse_optimizers.py:126-214): grow where the EMA of the
  dense gradients is largest."""

  def __init__(self, optimizer, begin_step, end_step, frequency,
               drop_fraction=0.1, drop_fraction_anneal='constant',
               use_locking=False, grow_init='zeros', momentum=0.9,
               use_tpu=False, name='SparseMomentumOptimizer',
               stateless_seed_offset=0, noise_std=1e-5):
    super().__init__(optimizer, begin_step, end_step, frequency,
                     drop_fraction=drop_fraction,
                     drop_fraction_anneal=drop_fraction_anneal,
                     grow_init=grow_init, use_locking=use_locking,
                     name='SparseMomentumOptimizer',
                     stateless_seed_offset=stateless_seed_offset,
                     noise_std=noise_std)
    del name
    self._ema_decay = float(momentum)
    self._use_tpu = use_tpu
    self._ema = {}
    self._masked_grads = None
    self._weight2masked_grads = {}

  def set_masked_grads(self, grads, weights):
    self._masked_grads = grads
    self._weight2masked_grads = {w.name: m for w, m in zip(weights, grads)}

  def compute_gradients(self, loss, **kwargs):
    grads_and_vars = self._optimizer.compute_gradients(loss, **kwargs)
    masked = self._optimizer.compute_gradients(
        loss, var_list=self.get_masked_weights())
    self.set_masked_grads([g for g, _ in masked], self.get_weights())
    return grads_and_vars

  def _before_apply_gradients(self, grads_and_vars):
    """tf.train.ExponentialMovingAverage.apply, zero-initialised shadows:
    ema -= (1 - decay) * (ema - g)   (:195-197)."""
    del grads_and_vars
    for w in self.get_weights():
      g = self._weight2masked_grads[w.name]
      ema = self._ema.get(w.name)
      if ema is None:
        ema = torch.zeros_like(g)
        self._ema[w.name] = ema
      ema.sub_((ema - g) * (1.0 - self._ema_decay))

  def ema_average(self, weights):
    return self._ema[weights.name]

  def _layer_request(self, lv, noise_std=None):
    noise_std = self._noise_std if noise_std is None else noise_std
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits,
               momentum=self._slot_of(lv),
               drop_noise=self._drop_noise(lv, noise_std),
               score_grow=self._ema[lv.weights.name].abs().contiguous().view(-1))
    self._attach_grow_values(req, lv)
    return req


def get_grow_grads(optimizer_or_graph=None):
  """The dense per-layer gradients RigL grows by, in ``get_weights()`` order --
  the values of the reference's ``_weight2masked_grads`` (the name
  comes from BASELINE.json's north star, the reference has no such symbol)."""
  if optimizer_or_graph is None:
    g = V.get_default_graph()
  elif isinstance(optimizer_or_graph, V.Graph):
    g = optimizer_or_graph
  else:
    opt = optimizer_or_graph
    w2g = getattr(opt, '_weight2masked_grads', None)
    if w2g:
      return [w2g[w.name] for w in opt.get_weights()]
    g = opt.graph
  return [w.grad for w in g.get_weights()]


class SparseSnipOptimizer(PruningGetterMixin, train.Optimizer):
  """SNIP (rigl/sparse_optimizers.py:217-338): at global_step 0, once, every
  mask becomes the top (n - floor(s*n)) entries of |g * w| (g = gradient of the
  still-dense layer); that call applies no gradients and does not advance the
  step.  Afterwards it is the plain inner optimizer.  Masks are derived by the
  K2 selection kernels (rigl_topk_mask_batched), all layers in one call."""

  def __init__(self, optimizer, default_sparsity, mask_init_method,
               custom_sparsity_map=None, use_locking=False, use_tpu=False,
               name='SparseSnipOptimizer'):
    super().__init__(use_locking, name, getattr(optimizer, '_graph', None))
    self._optimizer = optimizer
    self._use_tpu = use_tpu
    self._default_sparsity = default_sparsity
    self._mask_init_method = mask_init_method
    self._custom_sparsity_map = custom_sparsity_map or {}
    self.is_snipped = False

  def compute_gradients(self, loss, **kwargs):
    return self._optimizer.compute_gradients(loss, **kwargs)

  def get_slot_names(self):
    return self._optimizer.get_slot_names()

  def get_slot(self, var, name):
    return self._optimizer.get_slot(var, name)

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    """:258-338.  Returns True on the snip iteration."""
    from rigl_amd import sparse_utils  # pylint: disable=import-outside-toplevel
    gs = global_step if global_step is not None else \
        train.get_or_create_global_step(self.graph)
    if int(gs.value) == 0 and not self.is_snipped:
      masks = self.get_masks()
      sparsities = sparse_utils.get_sparsities(
          masks, self._mask_init_method, self._default_sparsity,
          self._custom_sparsity_map)
      items = []
      for l in self.graph.masked_layers():
        n = l.weights.numel
        n_keep = n - sparse_utils.get_n_zeros(n, sparsities[l.mask.name])
        # |g * v| (:301) with g the gradient of the WHOLE loss w.r.t. the raw variable: at step 0 the layer is still
        # dense, so g = dL/d(mask*W) + weight_decay * W -- the l2 term the update kernel otherwise folds in
        # (``rigl_masked_sgd_momentum``) belongs in the score (the reference's loss includes the regulariser,
        # imagenet_train_eval.py:578-584)
        # Under data parallelism the gradient arena holds the SUM over replicas (the 1 / world of the mean lives in the
        # update kernel's grad_scale), while every replica's loss carries the regulariser once: scale the summed data
        # gradient back to the mean before adding wd * W, or the l2 term would weigh 1 / world of what it does on one
        # GPU and the SNIP masks would depend on the number of GPUs.
        g_var = l.weights.grad
        sync = getattr(self._optimizer, '_grad_sync', None)
        scale = float(getattr(sync, 'grad_scale', 1.0)) if sync is not None else 1.0
        if scale != 1.0:
          g_var = g_var * scale
        if l.weights.weight_decay:
          g_var = g_var + float(l.weights.weight_decay) * l.weights.data
        score = (g_var * l.weights.data).abs().contiguous().view(-1)
        items.append((score, n_keep, l.mask.bits))
      ops.topk_mask_batched(items)
      self.graph.shadows_dirty = True
      self.is_snipped = True
      if hasattr(self._optimizer, '_backward_done_for'):
        self._optimizer._backward_done_for = None
      return True
    self._optimizer.apply_gradients(grads_and_vars, global_step=global_step,
                                    name=name)
    return False


class SparseDNWOptimizer(PruningGetterMixin, train.Optimizer):
  """Discovering Neural Wirings (rigl/sparse_optimizers.py:341-480): the DENSE
  gradient updates every weight (masked-out ones included), then every mask is
  re-derived each step as the top (n - floor(s*n)) entries of |w|."""

  def __init__(self, optimizer, default_sparsity, mask_init_method,
               custom_sparsity_map=None, use_tpu=False, use_locking=False,
               name='SparseDNWOptimizer'):
    super().__init__(use_locking, name, getattr(optimizer, '_graph', None))
    self._optimizer = optimizer
    self._use_tpu = use_tpu
    self._default_sparsity = default_sparsity
    self._mask_init_method = mask_init_method
    self._custom_sparsity_map = custom_sparsity_map or {}

  def compute_gradients(self, loss, var_list=None, **kwargs):
    # the gradient arena already holds d loss / d(mask*W): dense (:381-394)
    return self._optimizer.compute_gradients(loss, **kwargs)

  def get_slot_names(self):
    return self._optimizer.get_slot_names()

  def get_slot(self, var, name):
    return self._optimizer.get_slot(var, name)

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    from rigl_amd import sparse_utils  # pylint: disable=import-outside-toplevel
    # no `mask *` on the gradient, and NO l2 term on the masked kernels: the reference differentiates w.r.t. the
    # masked_weights tensors (:375-386), which the regulariser -- a function of the raw variables -- does not reach
    self._optimizer.dense_masked_update = True
    try:
      self._optimizer.apply_gradients(grads_and_vars, global_step=global_step,
                                      name=name)
    finally:
      self._optimizer.dense_masked_update = False
    masks = self.get_masks()
    sparsities = sparse_utils.get_sparsities(
        masks, self._mask_init_method, self._default_sparsity,
        self._custom_sparsity_map)
    items = []
    g = self.graph
    # |W| of the whole masked segment in one pass over the arena (one launch, not one per layer); a layer's scores are a
    # view of it.  Graphs that were not finalised (no arena) take the per-layer form.
    seg = g.seg.get(V.KIND_MASKED) if getattr(g, 'finalized', False) else None
    score = None
    if seg and seg[1] > seg[0]:
      # a persistent buffer (the selection runs every step: no 100 MB allocation per step)
      score = getattr(self, '_dnw_score', None)
      if score is None or score.numel() != seg[1] - seg[0] or score.device != g.W.device:
        score = self._dnw_score = torch.empty(seg[1] - seg[0], dtype=g.W.dtype, device=g.W.device)
      torch.abs(g.W[seg[0]:seg[1]], out=score)
    for l in g.masked_layers():
      n = l.weights.numel
      n_keep = n - sparse_utils.get_n_zeros(n, sparsities[l.mask.name])
      if score is not None and seg[0] <= l.weights.offset and l.weights.offset + n <= seg[1]:
        sc = score[l.weights.offset - seg[0]:l.weights.offset - seg[0] + n]
      else:
        sc = l.weights.data.abs().contiguous().view(-1)
      items.append((sc, n_keep, l.mask.bits))
    ops.topk_mask_batched(items)
    self.graph.shadows_dirty = True
    return None
