"""Variable store and HBM layout ("graph" of the eager twin).

The reference keeps its state in TF graph collections that
``tf.contrib.model_pruning`` fills (`masks`, `kernel`, `masked_weights`;
rigl/sparse_optimizers.py:46-56).  Here a ``Graph`` object plays that role and,
MI355X-first, owns the memory layout: after ``finalize()`` every variable is a
view into a few flat device arenas

    W      fp32   [ masked kernels | dense kernels | BN / bias ]   (master weights)
    G      fp32   same layout: dense gradients (what RigL's grow score reads,
                  what the DP all-reduce moves, what the update kernel eats)
    BITS   int32  1 bit / weight over the masked segment (bit = arena index)
    HWIO / OHWI   bf16 shadows of mask*W for the MFMA conv kernels

so that the masked SGD-momentum update is ONE launch over the masked segment,
the data-parallel all-reduce is a handful of large buckets of one buffer, and
the prune/regrow kernels walk all layers in the same launches.  Every tensor
starts on a 64-element (256 B) boundary, which keeps 16-B vector accesses
aligned and makes each layer's bitmap start on a word boundary.
"""
import re

import numpy as np
import torch

ALIGN = 64  # elements

KIND_MASKED = 0   # masked conv / fc kernel (has a mask, weight decay applies)
KIND_DENSE = 1    # unmasked conv / fc kernel (weight decay applies)
KIND_OTHER = 2    # BN scale/offset, biases (no mask; decay per variable)


def _align(n):
  return (int(n) + ALIGN - 1) // ALIGN * ALIGN


class Variable:
  """A named fp32 tensor (tf.Variable stand-in).  ``data`` is a private
  tensor until the graph is finalized, then a view into the W arena."""

  def __init__(self, graph, name, shape, kind, weight_decay=0.0, init=None,
               trainable=True):
    self.graph = graph
    self.name = name if name.endswith(':0') else name + ':0'
    self.shape = tuple(int(s) for s in shape)
    self.kind = kind
    self.weight_decay = float(weight_decay)
    self.trainable = trainable
    self.numel = int(np.prod(self.shape)) if self.shape else 1
    self.offset = None                       # element offset in the arenas
    self.data = torch.zeros(self.shape, dtype=torch.float32,
                            device=graph.device)
    if init is not None:
      self.assign(init)
    self.grad = torch.zeros(self.shape, dtype=torch.float32,
                            device=graph.device) if trainable else None
    self.initial_value = None

  @property
  def dtype(self):
    return torch.float32

  def assign(self, value):
    v = torch.as_tensor(np.asarray(value, dtype=np.float32) if not
                        torch.is_tensor(value) else value)
    self.data.copy_(v.to(self.data.device, torch.float32).reshape(self.shape))
    return self

  def numpy(self):
    return self.data.detach().cpu().numpy()

  def __repr__(self):
    return '<Variable %s %s>' % (self.name, self.shape)


class MaskVariable:
  """The `{scope}/mask:0` variable: stored as a bitmap (int32 words, 1 bit per
  weight, flat HWIO order); the 0/1 float view exists only on request."""

  def __init__(self, graph, name, shape):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    self.graph = graph
    self.name = name if name.endswith(':0') else name + ':0'
    self.shape = tuple(int(s) for s in shape)
    self.numel = int(np.prod(self.shape))
    self.offset = None
    self.bits = torch.full((ops.n_mask_words(self.numel),), -1,
                           dtype=torch.int32, device=graph.device)
    self._clear_tail()

  @property
  def dtype(self):
    return torch.float32

  def _clear_tail(self):
    rem = self.numel & 31
    if rem:
      self.bits[-1] = int((1 << rem) - 1)

  def assign(self, value):
    """value: 0/1 array or tensor of self.shape."""
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    v = value if torch.is_tensor(value) else torch.from_numpy(
        np.ascontiguousarray(np.asarray(value, dtype=np.float32)))
    v = v.to(self.bits.device, torch.float32).reshape(-1).contiguous()
    if v.numel() != self.numel:
      raise ValueError('mask %s: expected %d elements, got %d' %
                       (self.name, self.numel, v.numel()))
    packed = ops.mask_pack(v)
    self.bits[:packed.numel()].copy_(packed)
    self.graph.shadows_dirty = True
    return self

  @property
  def data(self):
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    return ops.mask_unpack(self.bits, self.shape)

  def numpy(self):
    return self.data.cpu().numpy()

  def sum(self):
    """Number of ones (popcount), as a python int."""
    w = self.bits.cpu().numpy().view(np.uint32)
    return int(np.unpackbits(w.view(np.uint8)).sum())

  def __repr__(self):
    return '<MaskVariable %s %s>' % (self.name, self.shape)


class MaskedLayerVars:
  """Everything one masked (or dense) conv/fc kernel owns."""

  def __init__(self, scope, weights, mask):
    self.scope = scope
    self.weights = weights
    self.mask = mask                  # None for dense ('baseline') layers
    self.hwio = None                  # bf16 [kh*kw*cin*cout] shadow of mask*W
    self.ohwi = None                  # bf16 transpose [cout][kh*kw*cin]
    self.k = int(np.prod(weights.shape[:-1]))
    self.cout = int(weights.shape[-1])

  @property
  def dense_grad(self):
    return self.weights.grad


class GlobalStep:
  """tf.train.get_or_create_global_step(): an int64 counter on the host.  The
  schedule (is_mask_update_iter / drop fraction) is scalar control flow; it
  never needs to touch the device."""

  def __init__(self):
    self.value = 0
    self.name = 'global_step:0'
    self.dtype = torch.int64

  def assign(self, v):
    self.value = int(v)

  def __int__(self):
    return int(self.value)

  def numpy(self):
    return np.int64(self.value)


class Graph:

  def __init__(self, device=None):
    if device is None:
      device = 'cuda:0' if torch.cuda.is_available() else 'cpu'
    self.device = torch.device(device)
    self.variables = {}          # name -> Variable / MaskVariable
    self.layers = []             # MaskedLayerVars in creation order
    self.layers_by_scope = {}
    self.modules = {}            # scope -> layer module (functional API reuse)
    self.global_step = None
    self.finalized = False
    self.shadows_dirty = True
    self._arena_vars = 0
    self.W = self.G = self.BITS = self.HWIO = self.OHWI = None
    self.seg = {}                # kind -> (begin, end) element range in W/G
    self._scope_counts = {}

  # ---- creation ------------------------------------------------------------
  def unique_scope(self, name, default):
    """tf.variable_scope(name, default_name): explicit names are used as is,
    default names get _1, _2 ... suffixes."""
    if name is not None:
      return name
    c = self._scope_counts.get(default, 0)
    self._scope_counts[default] = c + 1
    return default if c == 0 else '%s_%d' % (default, c)

  def add_variable(self, name, shape, kind, weight_decay=0.0, init=None,
                   trainable=True):
    full = name if name.endswith(':0') else name + ':0'
    if full in self.variables:
      raise ValueError('Variable %s already exists' % full)
    v = Variable(self, full, shape, kind, weight_decay, init, trainable)
    self.variables[full] = v
    self.finalized = False
    return v

  def add_masked_layer(self, scope, shape, masked, weight_decay=0.0, init=None):
    """Creates `{scope}/weights` (+ `{scope}/mask` when masked), registers them
    in the pruning collections in creation order."""
    w = self.add_variable(scope + '/weights', shape,
                          KIND_MASKED if masked else KIND_DENSE, weight_decay,
                          init)
    m = None
    if masked:
      m = MaskVariable(self, scope + '/mask', shape)
      self.variables[m.name] = m
    lv = MaskedLayerVars(scope, w, m)
    lv.hwio = torch.zeros(w.numel, dtype=torch.bfloat16, device=self.device)
    lv.ohwi = torch.zeros(w.numel, dtype=torch.bfloat16, device=self.device)
    self.layers.append(lv)
    self.layers_by_scope[scope] = lv
    self.shadows_dirty = True
    return lv

  def get_or_create_global_step(self):
    if self.global_step is None:
      self.global_step = GlobalStep()
    return self.global_step

  # ---- collections (tf.contrib.model_pruning.pruning.get_*) -----------------
  def masked_layers(self):
    return [l for l in self.layers if l.mask is not None]

  def get_weights(self):
    return [l.weights for l in self.masked_layers()]

  def get_masks(self):
    return [l.mask for l in self.masked_layers()]

  def trainable_variables(self):
    return [v for v in self.variables.values()
            if isinstance(v, Variable) and v.trainable]

  # ---- layout --------------------------------------------------------------
  def finalize(self):
    """(Re)builds the flat arenas and re-points every variable at its view.
    Idempotent; called again automatically when variables were added."""
    tv = self.trainable_variables()
    if self.finalized and self._arena_vars == len(tv):
      return
    order = ([v for v in tv if v.kind == KIND_MASKED] +
             [v for v in tv if v.kind == KIND_DENSE] +
             [v for v in tv if v.kind == KIND_OTHER])
    off = 0
    seg = {}
    for kind in (KIND_MASKED, KIND_DENSE, KIND_OTHER):
      begin = off
      for v in order:
        if v.kind == kind:
          v._new_offset = off
          off += _align(v.numel)
      seg[kind] = (begin, off)
    total = max(off, ALIGN)
    dev = self.device
    W = torch.zeros(total, dtype=torch.float32, device=dev)
    G = torch.zeros(total, dtype=torch.float32, device=dev)
    n_masked = seg[KIND_MASKED][1]
    BITS = torch.zeros(max(n_masked // 32, 1), dtype=torch.int32, device=dev)
    n_kernel = seg[KIND_DENSE][1]
    HWIO = torch.zeros(max(n_kernel, 1), dtype=torch.bfloat16, device=dev)
    OHWI = torch.zeros(max(n_kernel, 1), dtype=torch.bfloat16, device=dev)
    for v in order:
      o = v._new_offset
      view = W[o:o + v.numel].view(v.shape)
      with torch.no_grad():
        view.copy_(v.data.detach())
      v.data = view
      v._leaf = None
      gview = G[o:o + v.numel].view(v.shape)
      v.grad = gview
      v.offset = o
    for l in self.layers:
      o = l.weights.offset
      n = l.weights.numel
      if l.mask is not None:
        words = (n + 31) // 32
        bview = BITS[o // 32:o // 32 + words]
        bview.copy_(l.mask.bits[:words])
        l.mask.bits = bview
        l.mask.offset = o
      l.hwio = HWIO[o:o + n]
      l.ohwi = OHWI[o:o + n]
    self.W, self.G, self.BITS, self.HWIO, self.OHWI = W, G, BITS, HWIO, OHWI
    self.seg = seg
    self.finalized = True
    self._arena_vars = len(tv)
    self.shadows_dirty = True
    self._on_relayout()

  def _on_relayout(self):
    for cb in list(getattr(self, '_relayout_callbacks', [])):
      cb()

  def add_relayout_callback(self, cb):
    if not hasattr(self, '_relayout_callbacks'):
      self._relayout_callbacks = []
    self._relayout_callbacks.append(cb)

  def refresh_shadows(self, force=False):
    """bf16(mask*W) in HWIO and OHWI order for every conv/fc kernel: one
    batched launch (rigl_pack_weights_batched)."""
    from rigl_amd import ops  # pylint: disable=import-outside-toplevel
    if not (self.shadows_dirty or force):
      return
    sync = getattr(self, 'grad_sync', None)
    if sync is not None and not sync._state_synced:   # pylint: disable=protected-access
      sync.sync_initial_state()          # first forward of a data-parallel run: rank 0's weights / masks everywhere
    items = []
    for l in self.layers:
      items.append((l.weights.data.view(-1),
                    l.mask.bits if l.mask is not None else None, l.k, l.cout,
                    l.hwio, l.ohwi))
    ops.pack_weights_batched(items)
    self.shadows_dirty = False

  def zero_other_grads(self):
    """Zeroes the BN/bias gradient segment (autograd accumulates into it);
    kernel gradients are overwritten by the wgrad kernel every step."""
    if self.finalized:
      b, e = self.seg[KIND_OTHER]
      if e > b:
        self.G[b:e].zero_()
    else:
      for v in self.trainable_variables():
        if v.kind == KIND_OTHER:
          v.grad.zero_()

  def zero_grads(self):
    if self.finalized:
      self.G.zero_()
    else:
      for v in self.trainable_variables():
        v.grad.zero_()


_default_graph = None


def get_default_graph():
  global _default_graph
  if _default_graph is None:
    _default_graph = Graph()
  return _default_graph


def reset_default_graph(device=None):
  """tf.reset_default_graph()."""
  global _default_graph
  _default_graph = Graph(device)
  return _default_graph


def set_default_graph(g):
  global _default_graph
  _default_graph = g
  return g


def mask_extract_name_fn(mask_name):
  """rigl/sparse_utils.py:31-32."""
  return re.findall('(.+)/mask:0', mask_name)[0]
