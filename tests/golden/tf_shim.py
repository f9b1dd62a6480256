"""Minimal NumPy stand-in for the TensorFlow-1.x symbols that
``rigl/sparse_optimizers_base.py`` and ``rigl/sparse_utils.py`` import.

Purpose: TensorFlow cannot be installed in the build container (no network),
but the reference's hot-path logic is plain Python that *composes* a small set
of TF ops.  Installing this shim into ``sys.modules`` lets
``tests/golden/make_golden.py`` import and execute the reference's own code
(unmodified, from /root/reference) and record golden input/output vectors.
What stays "TF-documented, not executed" is only the semantics of the ops
implemented below -- chiefly ``nn_ops.top_k`` (value descending, equal values
lower index first) and ``cosine_decay`` (float32).

Execution model: eager, with *lazy* handling of exactly the things the
reference relies on graph semantics for:
  * ``Optimizer.apply_gradients`` of the inner optimizer returns a ``LazyOp``
    that runs only when it is a control dependency or the result of the taken
    ``cond`` branch (the reference builds one such op "to create slots" and
    never runs it -- sparse_optimizers_base.py:505-506).
  * ``control_flow_ops.cond`` evaluates the predicate, calls the taken branch
    and runs the op it returns.

TEST INFRASTRUCTURE ONLY.  Never imported by the product.
"""
import math
import sys
import types

import numpy as np


# ----------------------------------------------------------------------------
# tensors / variables
# ----------------------------------------------------------------------------
class _Shape(tuple):

  def as_list(self):
    return list(self)


def _val(x):
  if isinstance(x, Variable):
    return x.value
  return x


class Variable:
  """A named mutable array (tf.Variable)."""

  def __init__(self, value, name, dtype=None):
    self.value = np.array(value, dtype=dtype)
    self.name = name if name.endswith(':0') else name + ':0'
    self.initial_value = self.value.copy()

  @property
  def dtype(self):
    return self.value.dtype.type

  @property
  def shape(self):
    return _Shape(self.value.shape)

  def __array__(self, dtype=None, copy=None):
    return self.value if dtype is None else self.value.astype(dtype)

  def _bin(op):  # pylint: disable=no-self-argument
    def f(self, other):
      return op(self.value, _val(other))
    return f

  def _rbin(op):  # pylint: disable=no-self-argument
    def f(self, other):
      return op(_val(other), self.value)
    return f

  __add__ = _bin(np.add)
  __radd__ = _rbin(np.add)
  __sub__ = _bin(np.subtract)
  __rsub__ = _rbin(np.subtract)
  __mul__ = _bin(np.multiply)
  __rmul__ = _rbin(np.multiply)
  __truediv__ = _bin(np.true_divide)
  __rtruediv__ = _rbin(np.true_divide)
  __lt__ = _bin(np.less)
  __le__ = _bin(np.less_equal)
  __gt__ = _bin(np.greater)
  __ge__ = _bin(np.greater_equal)

  def __neg__(self):
    return -self.value


class LazyOp:
  """An op that has been built but not run."""

  def __init__(self, fn):
    self._fn = fn
    self._done = False

  def run(self):
    if not self._done:
      self._done = True
      self._fn()


def run_op(op):
  if isinstance(op, LazyOp):
    op.run()
  elif isinstance(op, (list, tuple)):
    for o in op:
      run_op(o)
  return op


class _VarStore:

  def __init__(self):
    self.vars = {}

  def reset(self):
    self.vars = {}


STORE = _VarStore()
# injected noise for (stateless_)random_normal: maps seed -> array, or None
NOISE_FOR_SEED = {}
UNIFORM_FOR_SEED = {}


# ----------------------------------------------------------------------------
# op modules
# ----------------------------------------------------------------------------
def _mod(name):
  m = types.ModuleType(name)
  m.__path__ = []
  sys.modules[name] = m
  parent, _, child = name.rpartition('.')
  if parent:
    setattr(sys.modules[parent], child, m)
  return m


def install():
  """Creates the fake ``tensorflow`` / ``google_research`` module trees."""
  for name in ['tensorflow', 'tensorflow.compat', 'tensorflow.python',
               'tensorflow.python.framework', 'tensorflow.python.ops',
               'tensorflow.python.tpu', 'tensorflow.python.tpu.ops',
               'tensorflow.python.training', 'google_research',
               'google_research.micronet_challenge']:
    _mod(name)

  # --- dtypes -------------------------------------------------------------
  dtypes = _mod('tensorflow.python.framework.dtypes')
  dtypes.float32 = np.float32
  dtypes.float64 = np.float64
  dtypes.int32 = np.int32
  dtypes.int64 = np.int64
  dtypes.bool = np.bool_

  # --- framework.ops ------------------------------------------------------
  fops = _mod('tensorflow.python.framework.ops')

  def convert_to_tensor(value, name=None, dtype=None):
    del name
    if isinstance(value, Variable):
      return value
    if dtype is None:
      if isinstance(value, float):
        dtype = np.float32
      elif isinstance(value, (int, np.integer)) and not isinstance(value, bool):
        dtype = np.int32
    return np.array(value, dtype=dtype)

  class control_dependencies:  # pylint: disable=invalid-name

    def __init__(self, deps):
      self.deps = deps

    def __enter__(self):
      run_op(list(self.deps))

    def __exit__(self, *a):
      return False

  fops.convert_to_tensor = convert_to_tensor
  fops.control_dependencies = control_dependencies

  # --- array_ops ----------------------------------------------------------
  aops = _mod('tensorflow.python.ops.array_ops')
  aops.size = lambda x: np.int32(np.asarray(x).size)
  aops.reshape = lambda x, shape: np.reshape(np.asarray(x), tuple(shape))
  aops.expand_dims = lambda x, axis: np.expand_dims(np.asarray(x), axis)
  aops.where = lambda c, a, b: np.where(np.asarray(c), np.asarray(a),
                                        np.asarray(b))
  aops.ones_like = lambda x, dtype=None: np.ones_like(np.asarray(x),
                                                      dtype=dtype)
  aops.zeros_like = lambda x, dtype=None: np.zeros_like(np.asarray(x),
                                                        dtype=dtype)
  aops.stack = lambda xs: np.stack([np.asarray(_val(x)) for x in xs])

  def scatter_nd(indices, updates, shape):
    out = np.zeros(tuple(shape), dtype=np.asarray(updates).dtype)
    idx = np.asarray(indices)[:, 0]
    # TF scatter_nd sums duplicates; indices here are a permutation.
    np.add.at(out, idx, np.asarray(updates))
    return out

  aops.scatter_nd = scatter_nd

  # --- math_ops -----------------------------------------------------------
  mops = _mod('tensorflow.python.ops.math_ops')

  def cast(x, dtype=None, name=None):
    del name
    x = np.asarray(_val(x))
    # TF float->int casts truncate toward zero, as numpy does.
    return x.astype(dtype)

  mops.cast = cast
  mops.abs = lambda x: np.abs(np.asarray(x))
  mops.sign = lambda x: np.sign(np.asarray(x))
  # TF reduces float32 in float32 (pairwise); exact for 0/1 data < 2**24.
  mops.reduce_sum = lambda x: np.sum(np.asarray(x), dtype=np.asarray(x).dtype)
  mops.reduce_min = lambda x: np.min(np.asarray(x))
  mops.reduce_mean = lambda x: np.mean(np.asarray(x),
                                       dtype=np.asarray(x).dtype)
  mops.reduce_std = lambda x: np.std(np.asarray(x))
  mops.range = lambda n: np.arange(int(n), dtype=np.int32)
  mops.equal = lambda a, b: np.equal(np.asarray(a), np.asarray(b))
  mops.logical_and = lambda a, b: np.logical_and(np.asarray(a), np.asarray(b))
  mops.logical_or = lambda a, b: np.logical_or(np.asarray(a), np.asarray(b))
  mops.greater_equal = lambda a, b: np.greater_equal(np.asarray(_val(a)),
                                                     np.asarray(_val(b)))
  mops.less_equal = lambda a, b: np.less_equal(np.asarray(_val(a)),
                                               np.asarray(_val(b)))
  mops.less = lambda a, b: np.less(np.asarray(_val(a)), np.asarray(_val(b)))
  mops.add = lambda a, b: np.add(np.asarray(_val(a)), np.asarray(_val(b)))
  mops.divide = lambda a, b: np.true_divide(np.asarray(a), np.asarray(b))
  mops.multiply = lambda a, b, name=None: np.multiply(np.asarray(a),
                                                      np.asarray(b))

  def _pow(x, y):
    x = np.asarray(x)
    return np.power(x, np.asarray(y, dtype=x.dtype))

  mops.pow = _pow

  # --- nn_ops.top_k -------------------------------------------------------
  nn = _mod('tensorflow.python.ops.nn_ops')

  def top_k(x, k):
    """TF TopK (CPU kernel, topk_op.cc): sorted by value descending; among
    equal values the LOWER index comes first (stable comparator)."""
    x = np.asarray(x)
    order = np.argsort(-x, kind='stable')[:int(k)].astype(np.int32)
    return x[order], order

  nn.top_k = top_k

  # --- control flow -------------------------------------------------------
  cf = _mod('tensorflow.python.ops.control_flow_ops')
  cf.no_op = lambda name=None: None

  def cond(pred, true_fn, false_fn):
    return run_op(true_fn() if bool(np.asarray(pred)) else false_fn())

  def Assert(condition, data):  # pylint: disable=invalid-name
    if not bool(np.asarray(condition)):
      raise AssertionError('tf.Assert failed: %s' % (data,))

  cf.cond = cond
  cf.Assert = Assert
  cf.group = lambda ops_: run_op(list(ops_))

  # --- init_ops / variable_scope -------------------------------------------
  iops = _mod('tensorflow.python.ops.init_ops')
  iops.constant_initializer = lambda value, dtype=None: (
      lambda shape, dt: np.full(tuple(shape), value, dtype=dt))
  iops.zeros_initializer = lambda: (lambda shape, dt: np.zeros(tuple(shape),
                                                               dtype=dt))
  vs = _mod('tensorflow.python.ops.variable_scope')

  def get_variable(name, shape=None, initializer=None, trainable=True,
                   dtype=np.float32):
    del trainable
    if name not in STORE.vars:
      if callable(initializer) and shape is None:
        value = initializer()
      else:
        value = initializer(shape, dtype)
      STORE.vars[name] = Variable(value, name, dtype=dtype)
    return STORE.vars[name]

  vs.get_variable = get_variable

  # --- state_ops ------------------------------------------------------------
  so = _mod('tensorflow.python.ops.state_ops')

  def assign(var, value, name=None):
    del name
    var.value = np.array(np.asarray(_val(value)), dtype=var.value.dtype).reshape(
        var.value.shape)
    return var

  so.assign = assign

  # --- random ops -----------------------------------------------------------
  def _lookup(table, seed, shape, dtype):
    key = int(np.asarray(seed).reshape(-1)[0]) if np.ndim(seed) else int(seed)
    if key in table:
      return np.asarray(table[key], dtype=dtype).reshape(tuple(shape))
    if None in table:
      return np.asarray(table[None](key, tuple(shape)), dtype=dtype)
    return np.zeros(tuple(shape), dtype=dtype)

  ro = _mod('tensorflow.python.ops.random_ops')
  sro = _mod('tensorflow.python.ops.stateless_random_ops')

  def random_normal(shape, mean=0., stddev=1., dtype=np.float32, seed=None):
    return _lookup(NOISE_FOR_SEED, seed, shape, dtype) * np.asarray(
        stddev, dtype=dtype) + np.asarray(mean, dtype=dtype)

  def random_uniform(shape, minval=0., maxval=1., dtype=np.float32, seed=None):
    u = _lookup(UNIFORM_FOR_SEED, seed, shape, dtype)
    return u * np.asarray(maxval - minval, dtype=dtype) + np.asarray(
        minval, dtype=dtype)

  ro.random_normal = random_normal
  ro.random_uniform = random_uniform
  ro.random_shuffle = lambda x: np.asarray(x)
  sro.stateless_random_normal = random_normal
  sro.stateless_random_uniform = random_uniform

  # --- tpu ops --------------------------------------------------------------
  tpu = _mod('tensorflow.python.tpu.ops.tpu_ops')
  tpu.cross_replica_sum = lambda g: g

  # --- learning_rate_decay.cosine_decay --------------------------------------
  lrd = _mod('tensorflow.python.training.learning_rate_decay')

  def cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0,
                   name=None):
    """TF 1.15 learning_rate_decay_v2.cosine_decay, in the dtype of
    ``learning_rate`` (float32)."""
    del name
    lr = np.asarray(_val(learning_rate))
    dt = lr.dtype.type
    gs = np.asarray(_val(global_step)).astype(dt)
    ds = np.asarray(_val(decay_steps)).astype(dt)
    gs = np.minimum(gs, ds)
    completed_fraction = dt(gs / ds)
    cosine_decayed = dt(dt(0.5) * dt(dt(1.0) + dt(np.cos(dt(dt(math.pi) * completed_fraction)))))
    decayed = dt(dt(dt(1) - dt(alpha)) * cosine_decayed + dt(alpha))
    return dt(lr * decayed)

  lrd.cosine_decay = cosine_decay

  # --- training_util ----------------------------------------------------------
  tu = _mod('tensorflow.python.training.training_util')

  def get_or_create_global_step():
    if 'global_step' not in STORE.vars:
      STORE.vars['global_step'] = Variable(np.int64(0), 'global_step',
                                           dtype=np.int64)
    return STORE.vars['global_step']

  tu.get_or_create_global_step = get_or_create_global_step

  # --- optimizer --------------------------------------------------------------
  opt = _mod('tensorflow.python.training.optimizer')

  class Optimizer:
    """tf.train.Optimizer surface used by the reference."""

    def __init__(self, use_locking, name):
      self._use_locking = use_locking
      self._name = name

    def minimize(self, loss, global_step=None, var_list=None):
      grads_and_vars = self.compute_gradients(loss, var_list=var_list) \
          if var_list is not None else self.compute_gradients(loss)
      return self.apply_gradients(grads_and_vars, global_step=global_step)

    def get_slot_names(self):
      return []

    def get_slot(self, var, name):
      return None

  opt.Optimizer = Optimizer

  class _Inner(Optimizer):
    """Inner optimizer whose gradients come from a user callback
    ``grad_fn(var_list_or_None) -> [(grad, var), ...]``."""

    def __init__(self, grad_fn, name):
      super().__init__(False, name)
      self._grad_fn = grad_fn
      self._slots = {}

    def compute_gradients(self, loss, var_list=None, **kwargs):
      del loss, kwargs
      return self._grad_fn(var_list)

    def _apply_one(self, g, v):
      raise NotImplementedError

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
      del name
      self._create_slots([v for _, v in grads_and_vars])
      # gradient VALUES are captured when the op is built (they are graph
      # tensors evaluated in the same session.run in TF).
      snapshot = [(np.array(np.asarray(g), copy=True), v)
                  for g, v in grads_and_vars if g is not None]

      def fn():
        for g, v in snapshot:
          self._apply_one(g, v)
        if global_step is not None:
          global_step.value = global_step.value + 1

      return LazyOp(fn)

    def _create_slots(self, var_list):
      pass

  class GradientDescentOptimizer(_Inner):

    def __init__(self, learning_rate, grad_fn):
      super().__init__(grad_fn, 'GradientDescent')
      self._lr = np.float32(learning_rate)

    def _apply_one(self, g, v):
      v.value = (v.value - (g.astype(np.float32) * self._lr)).astype(
          v.value.dtype)

  class MomentumOptimizer(_Inner):
    """tf.train.MomentumOptimizer; slot name 'momentum' (ApplyMomentum)."""

    def __init__(self, learning_rate, momentum, grad_fn, use_nesterov=False):
      super().__init__(grad_fn, 'Momentum')
      self._lr = np.float32(learning_rate)
      self._mu = np.float32(momentum)
      self._nesterov = use_nesterov

    def get_slot_names(self):
      return ['momentum']

    def get_slot(self, var, name):
      return self._slots[(var.name, name)]

    def _create_slots(self, var_list):
      for v in var_list:
        key = (v.name, 'momentum')
        if key not in self._slots:
          self._slots[key] = Variable(np.zeros_like(v.value),
                                      v.name[:-2] + '/Momentum')

    def _apply_one(self, g, v):
      acc = self._slots[(v.name, 'momentum')]
      g = g.astype(np.float32)
      a_new = ((acc.value * self._mu).astype(np.float32) + g).astype(np.float32)
      acc.value = a_new
      if self._nesterov:
        step = ((g * self._lr).astype(np.float32) +
                ((a_new * self._mu).astype(np.float32) * self._lr).astype(
                    np.float32)).astype(np.float32)
      else:
        step = (a_new * self._lr).astype(np.float32)
      v.value = (v.value - step).astype(np.float32)

  opt.GradientDescentOptimizer = GradientDescentOptimizer
  opt.MomentumOptimizer = MomentumOptimizer

  # --- tensorflow.compat.v1 (used by rigl/sparse_utils.py) --------------------
  v1 = _mod('tensorflow.compat.v1')
  logging = types.SimpleNamespace(info=lambda *a, **k: None)
  v1.logging = logging
  v1.float32 = np.float32
  v1.float64 = np.float64
  v1.int32 = np.int32
  v1.int64 = np.int64
  v1.constant = lambda value, dtype=None: np.array(value, dtype=dtype)
  v1.assign = assign
  v1.group = lambda ops_: list(ops_)
  v1.size = aops.size
  v1.cast = cast
  v1.reduce_sum = mops.reduce_sum
  sys.modules['tensorflow'].compat.v1 = v1

  counting = _mod('google_research.micronet_challenge.counting')
  del counting
  return sys.modules['tensorflow']
