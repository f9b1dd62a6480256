"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI via
``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" in CPU tests).

What the reference does: ``CrossShardOptimizer`` sums every
variable gradient over replicas with the loss pre-divided by the replica count
(imagenet_train_eval.py:363-365) and RigL sums the DENSE masked-weight
gradients (``cross_replica_sum``, sparse_optimizers_base.py:472-473) so every
replica derives the same mask with no mask traffic.

MI355X-first design: both exchanges are the SAME buffer here -- the dense
gradient arena ``Graph.G`` -- so one all-reduce per step serves the weight
update and (on update steps) the grow scores.  The arena is reduced in a few
large buckets (default 32 MB: xGMI is point-to-point, large messages amortise
per-link latency) launched from inside the backward pass as soon as the
layers of a bucket have produced their gradients (backward walks the arena
from its end to its start), so communication overlaps the remaining backward
kernels.  The mean (1/world) is folded into the update kernel's ``grad_scale``;
the mask update consumes the raw sum, like the reference.
"""
import os
import time

import torch
import torch.distributed as dist

DEBUG = os.environ.get('RIGL_DEBUG', '0') == '1'     # check_masks_identical after EVERY mask update (sparse_optimizers._run_update)


class GradSync:

  def __init__(self, graph, bucket_bytes=None, group=None, enabled=None):
    if bucket_bytes is None:
      bucket_bytes = int(float(os.environ.get('RIGL_DP_BUCKET_MB', '32')) * (1 << 20))
    self.graph = graph
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.enabled = (self.world > 1) if enabled is None else enabled
    # the mean over replicas rides in K3's grad_scale -- only when the sum is actually taken
    self.grad_scale = 1.0 / self.world if self.enabled else 1.0
    self._state_synced = False
    self.bucket_elems = max(int(bucket_bytes) // 4, 1)
    self._handles = []
    self._hi = None       # arena index above which everything is already sent
    self._vars = None     # kernel variables sorted by arena offset
    self._ptr = -1        # highest-offset variable not yet produced
    self._ready = set()
    self.n_buckets_last = 0
    # per-bucket record of the most recent step: (arena offset, elements, host ms since the step's first
    # notification, [device events]) -- bench.py prints it next to the stand-alone all-reduce figure
    self._timeline = []
    self._t_first = None
    self._timeline_events = False
    self._wait_events = None
    graph.grad_sync = self

  # ---- instrumentation ---------------------------------------------------------
  def reset_timeline(self, device_events=True):
    """Start recording: every bucket launch of the following steps is stamped (host clock; with
    ``device_events`` also a CUDA event on the compute stream at the launch point and one after the
    final wait, so that launch -> done spans are known on the device timeline)."""
    self._timeline = []
    self._timeline_events = bool(device_events) and self.graph.G.is_cuda

  def timeline_summary(self):
    """Buckets of the LAST step: where in the step each collective was launched and how long the
    compute stream then still had to wait for the exchange at the end of backward."""
    if not self._timeline:
      return None
    rows = []
    wait_ms = None
    if self._timeline_events and self._wait_events is not None:
      torch.cuda.synchronize()
      a, b = self._wait_events
      wait_ms = a.elapsed_time(b)
    first_ev = next((r[3] for r in self._timeline if r[3] is not None), None)
    for lo, n, t_ms, ev in self._timeline:
      row = {'arena_offset': lo, 'bytes': 4 * n, 'host_ms_after_first_ready': round(t_ms, 4)}
      if ev is not None and first_ev is not None:
        row['device_ms_after_first_bucket'] = round(first_ev.elapsed_time(ev), 4)
      rows.append(row)
    return {'buckets': rows, 'tail_wait_ms': wait_ms,
            'note': 'last timed step: one row per all-reduce launch in launch order (backward walks the arena end -> start; '
                    'the final row carries the head of the kernel segment together with the BN / bias segment); '
                    'tail_wait_ms = compute-stream time between the end of backward and the last bucket being done'}

  def sync_initial_state(self, src=0):
    """Replicas must start from the same weights, masks and BN buffers: the
    reference gets that from a shared initial checkpoint / identical seeds;
    here rank ``src`` broadcasts its arenas once (W, BITS and every BatchNorm's
    moving statistics), so a caller that seeded NumPy differently per rank (or not
    at all: ``get_mask_random`` draws from the global RNG) still trains ONE network."""
    if not self.enabled or self._state_synced:
      return
    g = self.graph
    g.finalize()
    dist.broadcast(g.W, src=src, group=self.group)
    if g.BITS is not None and g.BITS.numel():
      dist.broadcast(g.BITS, src=src, group=self.group)
    for mod in g.modules.values():
      for name in ('moving_mean', 'moving_variance'):
        t = getattr(mod, name, None)
        if torch.is_tensor(t):
          dist.broadcast(t, src=src, group=self.group)
    # per-layer initial values kept for grow_init='initial_dist' are part of the state a grown weight can take
    for v in g.trainable_variables():
      t = getattr(v, 'initial_value', None)
      if torch.is_tensor(t) and t.is_cuda:
        dist.broadcast(t, src=src, group=self.group)
    g.shadows_dirty = True
    self._state_synced = True
    # NOTE: this is a collective.  It runs at the first refresh_shadows() of a data-parallel run, i.e. the
    # first forward -- every rank must run that forward (call sync_initial_state() explicitly at the top of the training
    # loop when some ranks evaluate first).  Optimizer slots are created zero on every rank, so they need no broadcast.

  def _kernel_end(self):
    from rigl_amd import variables as V  # pylint: disable=import-outside-toplevel
    return self.graph.seg[V.KIND_DENSE][1] if self.graph.finalized else 0

  def _begin_step(self):
    from rigl_amd import variables as V  # pylint: disable=import-outside-toplevel
    self._vars = sorted((v for v in self.graph.trainable_variables()
                         if v.kind in (V.KIND_MASKED, V.KIND_DENSE)),
                        key=lambda v: v.offset)
    self._ptr = len(self._vars) - 1
    self._ready = set()
    self._hi = self._kernel_end()
    self._handles = []
    self._timeline = []
    self._t_first = time.perf_counter()

  def _reset(self):
    self._hi = None
    self._handles = []

  # called by the masked-layer autograd bridge right after wgrad is enqueued
  def notify_layer_grad_ready(self, var):
    """Backward produces kernel gradients from the end of the arena towards
    its start; a bucket is sent as soon as a contiguous tail of at least
    ``bucket_elems`` has been produced (out-of-order arrivals just wait)."""
    if not self.enabled:
      return
    if self._hi is None:
      self._begin_step()
    self._ready.add(var.offset)
    lo = None
    while self._ptr >= 0 and self._vars[self._ptr].offset in self._ready:
      lo = self._vars[self._ptr].offset
      self._ptr -= 1
    if lo is not None and self._hi - lo >= self.bucket_elems:
      self._launch(lo, self._hi)
      self._hi = lo

  def _stamp(self, lo, n):
    ev = None
    if self._timeline_events:
      ev = torch.cuda.Event(enable_timing=True)
      ev.record()
    t0 = self._t_first if self._t_first is not None else time.perf_counter()
    self._timeline.append((lo, n, (time.perf_counter() - t0) * 1e3, ev))

  def _launch(self, lo, hi):
    if hi > lo:
      self._stamp(lo, hi - lo)
      self._handles.append(dist.all_reduce(self.graph.G[lo:hi], group=self.group, async_op=True))

  def _launch_last(self, hi, kend):
    """The final flush: the head of the kernel segment [0, hi) and the BN / bias segment [kend, end) as ONE
    collective launch.  The BN / bias gradients are complete only when backward ends (the stem's batch norm
    is its last node), so they cannot ride in an earlier bucket; sent on their own they were a second,
    latency-bound all-reduce of ~0.2 MB behind the last bucket.  With RCCL both tensors go into one
    group call (torch's coalescing manager = ncclGroupStart / End: one kernel, one ring traversal);
    backends without coalescing (gloo in the CPU tests) get the two calls back to back."""
    g = self.graph
    end = g.G.numel()
    parts = [(lo, h) for lo, h in ((0, hi), (kend, end)) if h > lo]
    if not parts:
      return
    if len(parts) == 2 and g.G.is_cuda and dist.get_backend(self.group) == 'nccl':
      self._stamp(parts[0][0], sum(h - lo for lo, h in parts))
      try:
        with dist._coalescing_manager(group=self.group, device=g.G.device, async_ops=True) as cm:   # pylint: disable=protected-access
          for lo, h in parts:
            dist.all_reduce(g.G[lo:h], group=self.group)
        self._handles.append(cm)
        return
      except (AttributeError, TypeError, RuntimeError):
        self._timeline.pop()               # this torch has no usable coalescing manager: plain launches below
    for lo, h in parts:
      self._launch(lo, h)

  def all_reduce(self, graph=None):
    """Called once after backward: flushes what is left (the head of the kernel
    segment, the BN/bias segment) and makes the compute stream wait."""
    del graph
    if not self.enabled:
      self._reset()
      return
    g = self.graph
    kend = self._kernel_end()
    if self._hi is None:                                   # no layer notified (e.g. a model without masked convs)
      self._timeline = []
      self._t_first = time.perf_counter()
    hi = self._hi if self._hi is not None else kend
    self._launch_last(hi, kend)                           # remaining kernels + BN / bias segment, one launch
    self.n_buckets_last = len(self._handles)
    if self._timeline_events:
      a = torch.cuda.Event(enable_timing=True)
      a.record()
    for h in self._handles:
      h.wait()                                            # stream-level wait on GPU
    if self._timeline_events:
      b = torch.cuda.Event(enable_timing=True)
      b.record()
      self._wait_events = (a, b)
    self._reset()

  def check_masks_identical(self):
    """Debug guard: every rank must hold the same bitmap."""
    if self.world <= 1:
      return True
    bits = self.graph.BITS.to(torch.int64)
    chk = (bits * torch.arange(1, bits.numel() + 1, device=bits.device)).sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
    return bool((lo == hi).item())


def broadcast_drop_fraction(value, src=0, group=None):
  """The drop fraction is a pure function of global_step; this 4-byte broadcast
  is the safety net the north star asks for (rank 0 is authoritative)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return float(value)
  dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
  t = torch.tensor([float(value)], dtype=torch.float32, device=dev)
  dist.broadcast(t, src=src, group=group)
  return float(t.item())
