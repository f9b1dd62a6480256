"""TF2-style front-end over the same kernels: ``MaskUpdater`` / ``UpdateSchedule``
(rigl/rigl_tf2/mask_updaters.py:37-394).

Differences from the TF1 optimizers (rigl_amd/sparse_optimizers.py), all taken
from the reference:
  * the training step is always applied; the mask update is a separate call the
    training loop makes when ``schedule.is_update_iter(step)``
    (rigl_tf2/train.py:417-426): ``step % update_freq == 0`` and
    ``step <= last_update_step`` (``< 0``: forever, ``0``: never);
  * RigL's grow score is |grad| of ``loss_fn(val_x, val_y)`` on a held-out batch
    taken AFTER the step (mask_updaters.py:185-192), not a by-product of it;
  * drop noise is off by default (noise_std = 0), every optimizer slot of a new
    connection is reset to ZERO, grown weights start at zero;
  * ``prune_masks`` (one-shot magnitude pruning) keeps the top n_keep scores and
    grows nothing.

``model`` is a ``rigl_amd.variables.Graph`` (or anything with a ``.graph``);
``optimizer`` is one of rigl_amd.train's optimizers.  All layers of an update go
through one fused ``rigl_prune_regrow`` launch set (K2); explicit score tensors
use the kernel's score_drop / score_grow inputs.
"""
import math

import numpy as np
import torch

from rigl_amd import _lib
from rigl_amd import ops
from rigl_amd import pyhash
from rigl_amd import variables as V

F32 = np.float32


def _graph_of(model):
  if isinstance(model, V.Graph):
    return model
  g = getattr(model, 'graph', None)
  if isinstance(g, V.Graph):
    return g
  raise ValueError('model must be a rigl_amd Graph or carry one as .graph')


class MaskUpdater:
  """Base class for mask update algorithms (mask_updaters.py:37-192)."""

  def __init__(self, model, optimizer, use_stateless=True,
               stateless_seed_offset=0, loss_fn=None):
    self._model = model
    self._graph = _graph_of(model)
    self._optimizer = optimizer
    self._use_stateless = use_stateless
    self._stateless_seed_offset = stateless_seed_offset
    self._loss_fn = loss_fn
    self.val_x = self.val_y = None

  # ---- model introspection -------------------------------------------------------
  def get_all_pruning_layers(self):
    return list(self._graph.masked_layers())

  def get_vars_and_masks(self):
    layers = self.get_all_pruning_layers()
    return [l.mask for l in layers], [l.weights for l in layers]

  def _layer_of(self, mask, var):
    for l in self._graph.masked_layers():
      if l.mask is mask or l.weights is var:
        return l
    raise ValueError('no masked layer owns %r / %r' % (mask, var))

  # ---- scores (overridden) ---------------------------------------------------------
  def get_drop_scores(self, all_vars, all_masks):
    raise NotImplementedError

  def get_grow_scores(self, all_vars, all_masks):
    raise NotImplementedError

  # ---- updates -----------------------------------------------------------------------
  def prune_masks(self, prune_fraction):
    """Keeps the top (1 - prune_fraction) of each layer's active weights."""
    all_masks, all_vars = self.get_vars_and_masks()
    drop_scores = self.get_drop_scores(all_vars, all_masks)
    for mask, var, drop_score in zip(all_masks, all_vars, drop_scores):
      self.generic_mask_update(mask, var, drop_score, None, prune_fraction)

  def update_masks(self, drop_fraction):
    """Drops and regrows ``drop_fraction`` of every layer's connections -- all
    layers in one fused launch set."""
    all_masks, all_vars = self.get_vars_and_masks()
    drop_scores = self.get_drop_scores(all_vars, all_masks)
    grow_scores = self.get_grow_scores(all_vars, all_masks)
    reqs = [self._request(m, v, sd, sg)
            for m, v, sd, sg in zip(all_masks, all_vars, drop_scores, grow_scores)]
    self._run(reqs, drop_fraction, False)

  def generic_mask_update(self, mask, var, score_drop, score_grow,
                          drop_fraction, reinit_when_same=False):
    """Prunes (+ grows when ``score_grow`` is given) one layer; all tensors have
    the variable's shape (mask_updaters.py:100-159)."""
    if score_grow is None:
      self._graph.finalize()
      n_ones = int(mask.sum())
      n_prune = int(F32(n_ones) * F32(float(drop_fraction)))
      score = score_drop.detach().reshape(-1).float().contiguous()
      ops.topk_mask(score, n_ones - n_prune, out=mask.bits)
      self._graph.shadows_dirty = True
      return
    self._run([self._request(mask, var, score_drop, score_grow)], drop_fraction,
              reinit_when_same)

  def _request(self, mask, var, score_drop, score_grow):
    lv = self._layer_of(mask, var)
    self._graph.finalize()
    req = dict(w=lv.weights.data.view(-1), mask_bits=lv.mask.bits, dense_grad=None,
               momentum=None)
    if score_drop is not None:
      req['score_drop'] = score_drop.detach().reshape(-1).float().contiguous()
    if score_grow is not None:
      req['score_grow'] = score_grow.detach().reshape(-1).float().contiguous()
    else:
      req['dense_grad'] = lv.weights.grad.view(-1)        # kernel takes |dense grad| itself
    req['_lv'] = lv
    return req

  def _run(self, reqs, drop_fraction, reinit_when_same):
    # K2 resets ONE slot tensor per layer inside the launch; further slots (none of the
    # optimizers here has more than one) are reset from the new-connection set below.
    names = list(self._optimizer.get_slot_names())
    extra = []
    for r in reqs:
      lv = r.pop('_lv')
      if names:
        r['momentum'] = self._optimizer.get_slot(lv.weights, names[0]).view(-1)
      if len(names) > 1:
        extra.append((lv, ops.mask_unpack(lv.mask.bits, (lv.weights.numel,))))
    ops.prune_regrow(reqs, float(drop_fraction), grow_init_mode=_lib.GROW_ZEROS,
                     momentum_reset_mode=_lib.MOMRESET_ZEROS, initial_acc_scale=0.0,
                     reinit_when_same=reinit_when_same)
    for lv, old in extra:
      new = ops.mask_unpack(lv.mask.bits, (lv.weights.numel,))
      grown = (new > old) if not reinit_when_same else (new > 0)
      for s_name in names[1:]:
        self._optimizer.get_slot(lv.weights, s_name).view(-1)[grown] = 0
    self._graph.shadows_dirty = True

  def reset_momentum(self, var, new_connections):
    """mask_updaters.py:161-167 (the fused update does this for its own grown set)."""
    for s_name in self._optimizer.get_slot_names():
      slot = self._optimizer.get_slot(var, s_name)
      slot[new_connections.view_as(slot)] = 0

  # ---- randomness ----------------------------------------------------------------------
  def _iterations(self):
    it = getattr(self._optimizer, 'iterations', None)
    if it is not None:
      return int(it)
    return int(self._graph.get_or_create_global_step().value)

  def _random(self, dist, shape, seed, scale, shift):
    if not self._use_stateless:
      gen = torch.randn if dist == 'normal' else torch.rand
      return gen(tuple(shape), device=self._graph.device) * scale + shift
    n = int(np.prod(tuple(shape))) if len(tuple(shape)) else 1
    out = ops.stateless_random(n, self._stateless_seed_offset + int(seed), self._iterations(), dist,
                               scale=scale, shift=shift, device=self._graph.device)
    return out.view(tuple(shape))

  def _random_uniform(self, shape, minval=0., maxval=1., seed=0):
    return self._random('uniform', shape, seed, float(F32(maxval) - F32(minval)), float(minval))

  def _random_normal(self, shape, stddev=1.0, seed=0):
    return self._random('normal', shape, seed, float(stddev), 0.0)

  # ---- validation-batch gradients --------------------------------------------------------
  def set_validation_data(self, val_x, val_y):
    self.val_x, self.val_y = val_x, val_y

  def _get_gradients(self, all_vars):
    """Dense gradients of ``loss_fn(val_x, val_y)`` w.r.t. the masked kernels
    (summed over replicas when the optimizer carries a GradSync)."""
    if self._loss_fn is None:
      raise ValueError('this mask updater needs a loss_fn')
    g = self._graph
    g.finalize()
    g.zero_other_grads()
    loss = self._loss_fn(self.val_x, self.val_y)
    loss.backward()
    sync = getattr(self._optimizer, '_grad_sync', None)
    if sync is not None:
      sync.all_reduce(g)
    return [v.grad for v in all_vars]


def _magnitude_scores(updater, all_vars, all_masks, noise_std):
  out = []
  for mask, var in zip(all_masks, all_vars):
    score = (mask.data.view(var.shape) * var.data).abs()
    if noise_std != 0:
      score = score + updater._random_normal(score.shape, stddev=noise_std,
                                             seed=pyhash.name_hash(var.name + 'drop'))
    out.append(score)
  return out


class SET(MaskUpdater):
  """Magnitude drop, uniformly random grow (mask_updaters.py:195-217)."""

  def get_drop_scores(self, all_vars, all_masks, noise_std=0):
    return _magnitude_scores(self, all_vars, all_masks, noise_std)

  def get_grow_scores(self, all_vars, all_masks):
    return [self._random_uniform(var.shape, seed=pyhash.name_hash(var.name + 'grow'))
            for var in all_vars]


class RigL(MaskUpdater):
  """Magnitude drop, |validation gradient| grow (mask_updaters.py:220-237)."""

  def get_drop_scores(self, all_vars, all_masks, noise_std=0):
    return _magnitude_scores(self, all_vars, all_masks, noise_std)

  def get_grow_scores(self, all_vars, all_masks):
    return [g.abs() for g in self._get_gradients(all_vars)]


class RigLInverted(RigL):
  """Grows where the gradient is SMALLEST (mask_updaters.py:240-247)."""

  def get_grow_scores(self, all_vars, all_masks):
    return [-g.abs() for g in self._get_gradients(all_vars)]


class UpdateSchedule:
  """When and how much to update (mask_updaters.py:252-303)."""

  def __init__(self, mask_updater, init_drop_fraction, update_freq,
               last_update_step):
    self._mask_updater = mask_updater
    self.update_freq = update_freq
    self.last_update_step = last_update_step
    self.init_drop_fraction = F32(init_drop_fraction)
    self.last_drop_fraction = 0

  def get_drop_fraction(self, step):
    raise NotImplementedError

  def is_update_iter(self, step):
    if step < 0:
      raise AssertionError('step must be non-negative, got %r' % (step,))
    if self.last_update_step < 0:
      is_valid_step = True           # no last step
    elif self.last_update_step == 0:
      is_valid_step = False          # never update
    else:
      is_valid_step = step <= self.last_update_step
    return bool(is_valid_step and step % self.update_freq == 0)

  def update(self, step, check_update_iter=True):
    if check_update_iter and not self.is_update_iter(step):
      raise AssertionError('step %r is not a mask update step' % (step,))
    self.last_drop_fraction = self.get_drop_fraction(step)
    if self.last_drop_fraction > 0.:
      self._mask_updater.update_masks(self.last_drop_fraction)

  def prune(self, prune_fraction):
    self.last_drop_fraction = prune_fraction
    self._mask_updater.prune_masks(self.last_drop_fraction)

  def set_validation_data(self, val_x, val_y):
    self._mask_updater.set_validation_data(val_x, val_y)


class ConstantUpdateSchedule(UpdateSchedule):

  def get_drop_fraction(self, step):
    return self.init_drop_fraction


class CosineUpdateSchedule(UpdateSchedule):
  """tf.keras.experimental.CosineDecay(init, decay_steps=last_update_step,
  alpha=0) in fp32 (mask_updaters.py:313-326)."""

  def get_drop_fraction(self, step):
    decay_steps = F32(self.last_update_step)
    s = min(F32(step), decay_steps)
    completed = F32(s / decay_steps)
    cosine_decayed = F32(0.5) * (F32(1.0) + F32(math.cos(F32(math.pi) * completed)))
    return F32(self.init_drop_fraction * cosine_decayed)


class ScaledLRUpdateSchedule(UpdateSchedule):
  """drop_fraction = init_drop_fraction / lr(0) * lr(step) (mask_updaters.py:329-348)."""

  def __init__(self, mask_updater, init_drop_fraction, update_freq,
               last_update_step, optimizer):
    self._optimizer = optimizer
    self._initial_lr = self._get_lr(0)
    super().__init__(mask_updater, init_drop_fraction, update_freq,
                     last_update_step)

  def _get_lr(self, step):
    lr = getattr(self._optimizer, 'lr', None)
    if lr is None:
      lr = self._optimizer._lr           # rigl_amd.train optimizers
    return float(lr(step)) if callable(lr) else float(lr)

  def get_drop_fraction(self, step):
    return (self.init_drop_fraction / F32(self._initial_lr)) * F32(self._get_lr(step))


def get_mask_updater(model, optimizer, loss_fn, update_alg='', schedule_alg='lr',
                     update_freq=100, init_drop_fraction=0.3, last_update_step=-1,
                     use_stateless=True):
  """mask_updaters.py:351-394 (gin.configurable 'mask_updater')."""
  if not update_alg:
    return None
  elif update_alg == 'set':
    mask_updater = SET(model, optimizer, use_stateless=use_stateless)
  elif update_alg == 'rigl':
    mask_updater = RigL(model, optimizer, loss_fn=loss_fn, use_stateless=use_stateless)
  elif update_alg == 'rigl_inverted':
    mask_updater = RigLInverted(model, optimizer, loss_fn=loss_fn, use_stateless=use_stateless)
  else:
    raise ValueError('update_alg:%s  is not valid.' % update_alg)
  if schedule_alg == 'lr':
    return ScaledLRUpdateSchedule(mask_updater, init_drop_fraction, update_freq,
                                  last_update_step, optimizer)
  elif schedule_alg == 'cosine':
    return CosineUpdateSchedule(mask_updater, init_drop_fraction, update_freq,
                                last_update_step)
  elif schedule_alg == 'constant':
    return ConstantUpdateSchedule(mask_updater, init_drop_fraction, update_freq,
                                  last_update_step)
  raise ValueError('schedule_alg:%s  is not valid.' % schedule_alg)
