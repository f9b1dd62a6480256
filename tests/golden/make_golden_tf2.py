#!/usr/bin/env python3
"""Golden vectors for the TF2-style front-end, produced by EXECUTING the reference's
own rigl/rigl_tf2/mask_updaters.py (unmodified) over the NumPy TensorFlow shim.

Runs only where /root/reference exists (the build container); the fixtures it
writes (tf2_updater.npz, tf2_schedule.json) are committed.  TensorFlow's TF2
namespace (tf.math.top_k, tf.scatter_nd, tf.where, Variable.assign,
tf.keras.experimental.CosineDecay, ...) is provided by the same shim the TF1
goldens use; gin and rigl_tf2.utils (Keras pruning wrapper lookup) are stubbed --
the model is a list of fake pruning-wrapper layers.
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('RIGL_REFERENCE', '/root/reference')
sys.path.insert(0, HERE)

import tf_shim  # noqa: E402

tf_shim.install()
V = tf_shim.Variable


# ---- TF2 top-level namespace over the shim's op modules -----------------------------------
def install_tf2():
  tf = sys.modules['tensorflow']
  aops = sys.modules['tensorflow.python.ops.array_ops']
  mops = sys.modules['tensorflow.python.ops.math_ops']
  nn = sys.modules['tensorflow.python.ops.nn_ops']
  cf = sys.modules['tensorflow.python.ops.control_flow_ops']
  fops = sys.modules['tensorflow.python.framework.ops']
  lrd = sys.modules['tensorflow.python.training.learning_rate_decay']
  sro = sys.modules['tensorflow.python.ops.stateless_random_ops']
  ro = sys.modules['tensorflow.python.ops.random_ops']
  val = tf_shim._val
  tf.float32, tf.int32, tf.int64 = np.float32, np.int32, np.int64
  tf.Variable = V
  tf.size = aops.size
  tf.cast = mops.cast
  tf.reduce_sum = lambda x: mops.reduce_sum(np.asarray(val(x)))
  tf.reduce_min = lambda x: mops.reduce_min(np.asarray(val(x)))
  tf.reshape = lambda x, shape: aops.reshape(np.asarray(val(x)), shape)
  tf.expand_dims = aops.expand_dims
  tf.where = lambda c, a, b: aops.where(val(c), val(a), val(b))
  tf.range = mops.range
  tf.ones_like = lambda x, dtype=None: aops.ones_like(np.asarray(val(x)), dtype=dtype)
  tf.zeros_like = lambda x, dtype=None: aops.zeros_like(np.asarray(val(x)), dtype=dtype)
  tf.scatter_nd = aops.scatter_nd
  tf.stack = aops.stack
  tf.abs = lambda x: mops.abs(np.asarray(val(x)))
  tf.convert_to_tensor = fops.convert_to_tensor
  tf.logical_and = lambda a, b: mops.logical_and(val(a), val(b))
  tf.cond = lambda pred, t, f: (t() if bool(np.asarray(pred)) else f())
  m = types.ModuleType('tensorflow.math')
  m.top_k = lambda x, k: nn.top_k(np.asarray(val(x)), k)
  m.equal = lambda a, b: mops.equal(val(a), val(b))
  m.logical_and = tf.logical_and
  m.abs = tf.abs
  tf.math = m
  dbg = types.ModuleType('tensorflow.debugging')

  def assert_near(a, b):
    if abs(float(np.asarray(a)) - float(np.asarray(b))) > 1e-6:
      raise AssertionError('assert_near failed: %r vs %r' % (a, b))
  dbg.assert_near = assert_near
  dbg.Assert = cf.Assert
  tf.debugging = dbg
  rnd = types.ModuleType('tensorflow.random')
  rnd.stateless_uniform = sro.stateless_random_uniform
  rnd.stateless_normal = sro.stateless_random_normal
  rnd.uniform = ro.random_uniform
  rnd.normal = ro.random_normal
  tf.random = rnd
  keras = types.ModuleType('tensorflow.keras')
  exp = types.ModuleType('tensorflow.keras.experimental')

  class CosineDecay:
    """tf.keras.experimental.CosineDecay (TF 2.x learning_rate_schedule.py), float32."""

    def __init__(self, initial_learning_rate, decay_steps, alpha=0.0, name=None):
      self.init, self.decay_steps, self.alpha = initial_learning_rate, decay_steps, alpha

    def __call__(self, step):
      return lrd.cosine_decay(np.asarray(val(self.init), dtype=np.float32), step, self.decay_steps, self.alpha)
  exp.CosineDecay = CosineDecay
  keras.experimental = exp
  tf.keras = keras

  def assign(self, value):
    self.value = np.array(np.asarray(val(value)), dtype=self.value.dtype).reshape(self.value.shape)
    return self
  V.assign = assign
  V.numpy = lambda self: self.value

  gin = types.ModuleType('gin')
  gin.configurable = lambda *a, **k: (lambda fn: fn)
  sys.modules['gin'] = gin


install_tf2()
_pkg = types.ModuleType('rigl')
_pkg.__path__ = [os.path.join(REF, 'rigl')]
sys.modules['rigl'] = _pkg
_tf2 = types.ModuleType('rigl.rigl_tf2')
_tf2.__path__ = [os.path.join(REF, 'rigl', 'rigl_tf2')]
sys.modules['rigl.rigl_tf2'] = _tf2
_utils = types.ModuleType('rigl.rigl_tf2.utils')


class PruningWrapper:            # stands in for tfmot's PruneLowMagnitude

  def __init__(self, var, mask):
    self.trainable = True
    self.pruning_vars = [(var, mask, None)]


_utils.PRUNING_WRAPPER = PruningWrapper
sys.modules['rigl.rigl_tf2.utils'] = _utils
_tf2.utils = _utils

from rigl.rigl_tf2 import mask_updaters as ref_mu  # noqa: E402


class Model:

  def __init__(self, layers):
    self.layers = layers


class Opt:
  """Optimizer surface mask_updaters touches: iterations, lr, slots."""

  def __init__(self, vars_, slot_values, lr=0.1):
    self.iterations = np.int64(40)
    self.lr = lr
    self._slots = {id(v): V(s, 'slot', dtype=np.float32) for v, s in zip(vars_, slot_values)}

  def get_slot_names(self):
    return ['momentum']

  def get_slot(self, var, name):
    return self._slots[id(var)]


def build(rng, shapes, sparsity=0.7):
  vars_, masks, slots, grads = [], [], [], []
  for i, sh in enumerate(shapes):
    w = rng.standard_normal(sh).astype(np.float32)
    m = (rng.random(sh) >= sparsity).astype(np.float32)
    if i == 0:                      # ties: duplicated magnitudes and an active exact zero
      w.reshape(-1)[5:9] = w.reshape(-1)[4]
      m.reshape(-1)[4:9] = 1.0
      w.reshape(-1)[11] = 0.0
      m.reshape(-1)[11] = 1.0
    vars_.append(V(w, 'layer%d/kernel' % i, dtype=np.float32))
    masks.append(V(m, 'layer%d/mask' % i, dtype=np.float32))
    slots.append(rng.standard_normal(sh).astype(np.float32))
    g = rng.standard_normal(sh).astype(np.float32)
    if i == 0:
      g.reshape(-1)[20:24] = g.reshape(-1)[19]
    grads.append(g)
  return vars_, masks, slots, grads


def gen_updates():
  out = {}
  shapes = [(24, 16), (3, 3, 8, 16), (64,)]
  for alg in ('rigl', 'rigl_inverted', 'set'):
    for frac in (0.3, 0.0, 1.0):
      rng = np.random.default_rng({'rigl': 1, 'rigl_inverted': 2, 'set': 3}[alg] * 10 + int(frac * 10))
      vars_, masks, slots, grads = build(rng, shapes)
      layers = [PruningWrapper(v, m) for v, m in zip(vars_, masks)]
      opt = Opt(vars_, slots)
      tag = '%s_f%02d' % (alg, int(frac * 10))
      for i in range(len(shapes)):
        out['%s__w%d' % (tag, i)] = vars_[i].value.copy()
        out['%s__m%d' % (tag, i)] = masks[i].value.copy()
        out['%s__a%d' % (tag, i)] = slots[i].copy()
        out['%s__g%d' % (tag, i)] = grads[i].copy()
      if alg == 'set':
        up = ref_mu.SET(Model(layers), opt, use_stateless=True)
        uni = [rng.random(sh).astype(np.float32) for sh in shapes]
        # SET's scores come from stateless_uniform(seed=hash(name+'grow')) -- salted per process -- so the
        # draws are injected at get_grow_scores; everything downstream is the reference's own code
        up.get_grow_scores = lambda all_vars, all_masks, _u=uni: list(_u)
        for i, u in enumerate(uni):
          out['%s__u%d' % (tag, i)] = u
      else:
        cls = ref_mu.RigL if alg == 'rigl' else ref_mu.RigLInverted
        up = cls(Model(layers), opt, loss_fn=None, use_stateless=True)
        up._get_gradients = lambda all_vars, _g=grads: list(_g)       # the validation-batch gradients
      up.update_masks(np.float32(frac))
      for i in range(len(shapes)):
        out['%s__w%d_new' % (tag, i)] = vars_[i].value.copy()
        out['%s__m%d_new' % (tag, i)] = masks[i].value.copy()
        out['%s__a%d_new' % (tag, i)] = opt.get_slot(vars_[i], 'momentum').value.copy()
  # prune_masks and generic_mask_update(reinit_when_same=True)
  rng = np.random.default_rng(77)
  vars_, masks, slots, grads = build(rng, shapes)
  layers = [PruningWrapper(v, m) for v, m in zip(vars_, masks)]
  opt = Opt(vars_, slots)
  up = ref_mu.RigL(Model(layers), opt, loss_fn=None)
  for i in range(len(shapes)):
    out['prune__w%d' % i] = vars_[i].value.copy()
    out['prune__m%d' % i] = masks[i].value.copy()
  up.prune_masks(np.float32(0.4))
  for i in range(len(shapes)):
    out['prune__m%d_new' % i] = masks[i].value.copy()
    out['prune__w%d_new' % i] = vars_[i].value.copy()
  rng = np.random.default_rng(78)
  vars_, masks, slots, grads = build(rng, shapes[:1])
  opt = Opt(vars_, slots)
  up = ref_mu.MaskUpdater(Model([PruningWrapper(vars_[0], masks[0])]), opt)
  sd = rng.random(shapes[0]).astype(np.float32)
  sg = rng.random(shapes[0]).astype(np.float32)
  out['reinit__w'], out['reinit__m'], out['reinit__a'] = vars_[0].value.copy(), masks[0].value.copy(), slots[0].copy()
  out['reinit__sd'], out['reinit__sg'] = sd, sg
  up.generic_mask_update(masks[0], vars_[0], sd, sg, np.float32(0.5), reinit_when_same=True)
  out['reinit__w_new'], out['reinit__m_new'] = vars_[0].value.copy(), masks[0].value.copy()
  out['reinit__a_new'] = opt.get_slot(vars_[0], 'momentum').value.copy()
  return out


def gen_schedules():
  class U:
    def update_masks(self, f):
      pass

    def prune_masks(self, f):
      pass
  out = {'gate': [], 'cosine': [], 'lr': [], 'constant': []}
  for last in (-1, 0, 250, 1000):
    s = ref_mu.ConstantUpdateSchedule(U(), 0.3, 100, last)
    out['gate'].append({'last_update_step': last, 'update_freq': 100,
                        'update_steps': [k for k in range(0, 1201, 50) if bool(s.is_update_iter(k))]})
  c = ref_mu.CosineUpdateSchedule(U(), 0.3, 100, 1000)
  for step in (0, 1, 100, 333, 500, 999, 1000, 1500):
    out['cosine'].append([step, float(np.float32(c.get_drop_fraction(step))).hex()])

  class O:
    def __init__(self, lr):
      self.lr = lr
  sched = ref_mu.ScaledLRUpdateSchedule(U(), 0.3, 10, -1, O(lambda step: 0.1 * (0.5 ** (step // 100))))
  for step in (0, 99, 100, 250, 777):
    out['lr'].append([step, float(np.float32(sched.get_drop_fraction(step))).hex()])
  k = ref_mu.ConstantUpdateSchedule(U(), 0.2, 10, -1)
  out['constant'] = [float(np.float32(k.get_drop_fraction(s_))).hex() for s_ in (0, 10, 12345)]
  return out


def main():
  np.savez_compressed(os.path.join(HERE, 'tf2_updater.npz'), **gen_updates())
  with open(os.path.join(HERE, 'tf2_schedule.json'), 'w') as f:
    json.dump(gen_schedules(), f, indent=1)
  print('TF2 front-end golden vectors written to', HERE)


if __name__ == '__main__':
  main()
