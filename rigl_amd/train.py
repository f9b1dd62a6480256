"""Inner optimizers with the tf.train.Optimizer surface the reference wraps
(``compute_gradients`` / ``apply_gradients`` / ``minimize`` / slots), executing
on the fused HIP update kernel (K3, rigl_masked_sgd_momentum).

  tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True)
      rigl/imagenet_resnet/imagenet_train_eval.py:360-361,
      rigl/cifar_resnet/resnet_train_eval.py:202-203, mnist_train_eval.py:263
  tf.train.GradientDescentOptimizer(lr)      rigl/sparse_optimizers_test.py:44

Eager twin of the TF1 graph API: ``compute_gradients(loss)`` runs the backward
pass (the masked-layer autograd bridge deposits DENSE kernel gradients in the
gradient arena); ``apply_gradients`` enqueues one update launch per arena
segment:  masked kernels (bitmap, weight decay) | dense kernels | BN & biases.
The l2 regulariser's gradient (scale * w on the raw variable) is
applied inside the kernel rather than through the loss.
"""
import torch

from rigl_amd import ops
from rigl_amd import variables as V


def get_or_create_global_step(graph=None):
  return (graph or V.get_default_graph()).get_or_create_global_step()


def _lr_value(lr, global_step):
  if callable(lr):
    return float(lr(int(global_step.value) if global_step is not None else 0))
  return float(lr)


class Optimizer:
  """Minimal tf.train.Optimizer base."""

  def __init__(self, use_locking=False, name='Optimizer', graph=None):
    self._use_locking = use_locking
    self._name = name
    self._graph = graph

  @property
  def graph(self):
    return self._graph or V.get_default_graph()

  def get_slot_names(self):
    return []

  def get_slot(self, var, name):
    raise KeyError(name)

  def minimize(self, loss, global_step=None, var_list=None, **kwargs):
    grads_and_vars = self.compute_gradients(loss, var_list=var_list, **kwargs) \
        if var_list is not None else self.compute_gradients(loss, **kwargs)
    return self.apply_gradients(grads_and_vars, global_step=global_step)


class GradientDescentOptimizer(Optimizer):
  """w -= lr * g  (g = mask * dense_grad + wd * w for masked kernels)."""

  def __init__(self, learning_rate, use_locking=False,
               name='GradientDescent', graph=None, grad_sync=None):
    super().__init__(use_locking, name, graph)
    self._lr = learning_rate
    self._momentum = None
    self._nesterov = False
    self._slot = None              # flat momentum arena (same layout as W)
    self._grad_sync = grad_sync    # rigl_amd.dist.GradSync or None
    self._backward_done_for = None
    self.dense_masked_update = False   # DNW: apply the dense gradient to masked kernels too

  # ---- gradients -------------------------------------------------------------
  def compute_gradients(self, loss, var_list=None, **kwargs):
    """Returns [(grad, var)].  With ``var_list`` = the masked weights (RigL's
    second call, sparse_optimizers_base.py:481-482) the DENSE gradients are
    returned -- they are a by-product of the same backward pass."""
    del kwargs
    g = self.graph
    g.finalize()
    if loss is not None and self._backward_done_for is not loss:
      g.zero_other_grads()
      loss.backward()
      self._backward_done_for = loss
      if self._grad_sync is not None:
        self._grad_sync.all_reduce(g)
    if var_list is not None:
      return [(v.grad, v) for v in var_list]
    # (the list handed out is remembered: getting it back untouched needs no per-variable check in apply_gradients)
    self._last_gv = [(v.grad, v) for v in g.trainable_variables()]
    return self._last_gv

  # ---- update ----------------------------------------------------------------
  def _ensure_slots(self):
    pass

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    del name
    g = self.graph
    g.finalize()
    # The kernels read the gradient arena.  Gradients handed back unchanged ARE arena slices; transformed ones
    # (clipping, scaling) are copied into their variable's slice; what the arena-wide update cannot express --
    # a None gradient or a subset of the trainable variables -- raises instead of being silently ignored.
    if grads_and_vars is not None and grads_and_vars is not getattr(self, '_last_gv', None):
      gv = list(grads_and_vars)
      seen = set()
      for grad, var in gv:
        if grad is None:
          raise NotImplementedError('apply_gradients: None gradient for %s (the fused update walks whole arena '
                                    'segments; freeze a variable by zeroing its gradient instead)' % var.name)
        if not hasattr(var, 'grad') or var.grad is None:
          raise ValueError('apply_gradients: %r is not a variable of this graph' % (var,))
        if grad.data_ptr() != var.grad.data_ptr():
          var.grad.copy_(grad.reshape(var.grad.shape))
        seen.add(var.name)
      missing = [v.name for v in g.trainable_variables() if v.name not in seen]
      if gv and missing:
        raise NotImplementedError('apply_gradients: the fused update covers every trainable variable; missing %s%s'
                                  % (missing[:3], ' ...' if len(missing) > 3 else ''))
    self._ensure_slots()
    lr = _lr_value(self._lr, global_step)
    scale = self._grad_sync.grad_scale if self._grad_sync is not None else 1.0
    mu = float(self._momentum) if self._momentum is not None else 0.0
    # one launch per segment; weight decay is uniform inside a kernel segment
    for kind in (V.KIND_MASKED, V.KIND_DENSE):
      b, e = g.seg[kind]
      if e <= b:
        continue
      wds = {v.weight_decay for v in g.trainable_variables() if v.kind == kind}
      if len(wds) > 1:
        self._apply_per_variable(kind, lr, mu, scale)
        continue
      wd = wds.pop() if wds else 0.0
      if kind == V.KIND_MASKED and self.dense_masked_update:
        wd = 0.0                 # DNW: gradients are taken w.r.t. mask*W, which the l2 regulariser does not reach
      ops.masked_sgd_momentum(
          g.W[b:e], g.G[b:e], lr,
          momentum=self._slot[b:e] if self._slot is not None else None,
          mask_bits=(g.BITS[b // 32:e // 32]
                     if kind == V.KIND_MASKED and not self.dense_masked_update else None),
          mu=mu, weight_decay=wd, grad_scale=scale, nesterov=self._nesterov)
    b, e = g.seg[V.KIND_OTHER]
    if e > b:
      ops.masked_sgd_momentum(
          g.W[b:e], g.G[b:e], lr,
          momentum=self._slot[b:e] if self._slot is not None else None,
          mu=mu, weight_decay=0.0, grad_scale=scale, nesterov=self._nesterov)
    g.shadows_dirty = True
    self._backward_done_for = None
    if global_step is not None:
      global_step.value += 1
    return None

  def _apply_per_variable(self, kind, lr, mu, scale):
    g = self.graph
    for l in g.layers:
      v = l.weights
      if v.kind != kind:
        continue
      o, n = v.offset, v.numel
      n4 = (n + 3) // 4 * 4  # padding inside the 64-aligned slot is harmless
      ops.masked_sgd_momentum(
          g.W[o:o + n4], g.G[o:o + n4], lr,
          momentum=self._slot[o:o + n4] if self._slot is not None else None,
          mask_bits=(l.mask.bits if l.mask is not None and not self.dense_masked_update else None), mu=mu,
          weight_decay=(0.0 if (kind == V.KIND_MASKED and self.dense_masked_update) else v.weight_decay), grad_scale=scale,
          nesterov=self._nesterov)


class MomentumOptimizer(GradientDescentOptimizer):
  """TF ApplyMomentum: accum = accum*momentum + g;
  w -= lr*g + lr*momentum*accum (use_nesterov) | w -= lr*accum."""

  def __init__(self, learning_rate, momentum, use_locking=False,
               name='Momentum', use_nesterov=False, graph=None, grad_sync=None):
    super().__init__(learning_rate, use_locking, name, graph, grad_sync)
    self._momentum = momentum
    self._nesterov = use_nesterov

  def get_slot_names(self):
    return ['momentum']

  def _ensure_slots(self):
    g = self.graph
    if self._slot is None or self._slot.numel() != g.W.numel():
      old = self._slot
      self._slot = torch.zeros_like(g.W)
      if old is not None:
        raise RuntimeError('variables were added after the momentum slots '
                           'were created')

  def get_slot(self, var, name):
    if name != 'momentum':
      raise KeyError(name)
    self.graph.finalize()
    self._ensure_slots()
    return self._slot[var.offset:var.offset + var.numel].view(var.shape)


class GraphedStep:
  """One training step replayed from a captured HIP graph (torch.cuda.CUDAGraph over the kernels the
  C ABI enqueues on the current stream) -- for launch-bound models, where the ~250 launches of a step cost
  more on the host than on the device (CIFAR WRN-22 at batch 128: 2.4 ms eager, the device work is a
  fraction of that).  ResNet-50 at batch 128 is device-bound and gains < 1 % (tools/graph_probe.py), so the
  headline benchmark stays eager.

  Only ORDINARY iterations are replayed: the schedule (``is_mask_update_iter``) is host arithmetic on the
  global step, so it is evaluated before every call and mask-update iterations (one in ``frequency``) run
  eagerly -- they change the masks the captured kernels read through the same buffers, which is fine: the
  graph holds pointers, not values.  Scalars that are kernel ARGUMENTS are frozen at capture: the learning
  rate (a new graph is captured whenever ``lr(global_step)`` changes: piecewise-constant schedules recapture
  a handful of times) and the gradient scale.  The loss tensor returned is the captured step's output buffer.

  ``loss_fn`` must read its inputs from fixed device buffers (copy each new batch into them).  Data-parallel
  steps (a GradSync with world > 1) are not captured -- the collectives stay eager -- and run as usual.
  """

  def __init__(self, loss_fn, optimizer, global_step, warmup=3):
    self._loss_fn = loss_fn
    self._opt = optimizer
    self._gs = global_step
    self._warmup = warmup
    self._graphs = {}          # lr value -> (CUDAGraph, loss tensor)
    inner = getattr(optimizer, '_optimizer', optimizer)
    self._inner = inner
    sync = getattr(inner, '_grad_sync', None)
    self._eager_only = (sync is not None and getattr(sync, 'enabled', False)) or not torch.cuda.is_available()
    self.replays = 0
    self.eager_steps = 0
    self._ws_gen = ops.WORKSPACE_GENERATION
    self._last_key = None      # learning-rate key of the previous ordinary step and how many steps it has been stable
    self._stable = 0
    self._stable_needed = 1    # capture at the second consecutive ordinary step with the same learning rate

  def _is_update(self):
    o = self._opt
    if not hasattr(o, 'is_mask_update_iter'):
      return False
    return bool(o.is_mask_update_iter(int(self._gs.value), o._last_update_step))   # pylint: disable=protected-access

  def _eager(self):
    loss = self._loss_fn()
    self._opt.minimize(loss, self._gs)
    self.eager_steps += 1
    return loss

  def __call__(self):
    if self._eager_only or self._is_update():
      return self._eager()
    key = _lr_value(self._inner._lr, self._gs)                   # pylint: disable=protected-access
    if self._ws_gen != ops.WORKSPACE_GENERATION:
      # a scratch buffer was (re)allocated since the graphs were captured (an eager mask update outgrew one, typically):
      # EVERY cached graph may point at the old buffer -- drop them all and capture again.  Checked before every replay,
      # so no graph is ever replayed against a freed buffer (ops.workspace keeps nothing alive for them).
      self._graphs.clear()
      self._ws_gen = ops.WORKSPACE_GENERATION
    ent = self._graphs.get(key)
    if ent is None:
      #  a learning rate that changes every step (linear warm-up, cosine) never reuses a graph keyed on its
      # value: while the key keeps changing, run plain eager steps and do not capture; once it has been stable for
      # `_stable_needed` consecutive steps (the constant plateaus of a piecewise schedule), capture again.
      if key == self._last_key:
        self._stable += 1
      else:
        self._stable = 0
        self._last_key = key
      if self._stable < self._stable_needed:
        return self._eager()
      if self._warmup > 0:                                         # allocator / workspace caches / descriptors settle first
        self._warmup -= 1
        return self._eager()
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        out = self._eager()                                        # THIS call's step, on a side stream as torch's capture recipe asks
      torch.cuda.current_stream().wait_stream(side)
      torch.cuda.synchronize()
      if self._is_update() or _lr_value(self._inner._lr, self._gs) != key:   # pylint: disable=protected-access
        return out                                                 # the next iteration is not an ordinary one at this lr: capture later
      gen0 = ops.WORKSPACE_GENERATION
      step1 = self._gs.value
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph):
        loss = self._loss_fn()
        self._opt.minimize(loss, self._gs)                         # capture enqueues nothing; undo its host side effect:
      self._gs.value = step1
      if gen0 != ops.WORKSPACE_GENERATION or gen0 != self._ws_gen:
        # a buffer grew during the pre-capture step or the capture itself: graphs cached under other learning rates are
        # stale, and this one may hold both addresses -- keep none, the next call captures on settled buffers
        self._graphs.clear()
        self._ws_gen = ops.WORKSPACE_GENERATION
        return out
      if len(self._graphs) >= 16:
        self._graphs.pop(next(iter(self._graphs)))                 # bound the cache (piecewise-constant schedules have few values)
      self._graphs[key] = (graph, loss)
      return out
    graph, loss = ent
    self._last_key, self._stable = key, self._stable_needed
    graph.replay()
    self._gs.value += 1
    self._inner.graph.shadows_dirty = True                         # the replayed update rewrote the weights
    self.replays += 1
    return loss
