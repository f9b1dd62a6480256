"""Mask initialisation and per-layer sparsity distributions with the API of
rigl/sparse_utils.py (``get_mask_random``, ``get_sparsities``,
``get_mask_init_fn`` ...).

Host-side NumPy, as in the reference: mask init happens once, before training
(`Scaffold.init_fn`, imagenet_train_eval.py:637-653); the result is uploaded
into the graph's bitmap arena.  The arithmetic order of the Erdos-Renyi solve
is kept identical to the reference (float64, same summation order), so the
sparsities are bit-identical -- checked against golden values produced by the
reference's own code (tests/golden/sparsities.json).
"""
import re

import numpy as np

DEFAULT_ERK_SCALE = 1.0


def mask_extract_name_fn(mask_name):
  """'{scope}/mask:0' -> '{scope}'  (sparse_utils.py:31-32)."""
  return re.findall('(.+)/mask:0', mask_name)[0]


def get_n_zeros(size, sparsity):
  """floor(sparsity * size)  (sparse_utils.py:35-36)."""
  return int(np.floor(sparsity * size))


def calculate_sparsity(masks):
  """1 - sum(ones) / sum(sizes) over mask variables (sparse_utils.py:39-45)."""
  dense = 0.
  ones = 0.
  for m in masks:
    dense += float(m.numel)
    ones += float(m.sum())
  return 1. - ones / dense


def get_mask_random_numpy(mask_shape, sparsity, random_state=None):
  """A 0/1 array with exactly floor(sparsity*n) zeros at shuffled positions
  (sparse_utils.py:48-68): zeros first, then ONE shuffle of the flat array --
  with ``random_state`` or the global NumPy RNG, whose consumption order across
  layers is part of reproducibility."""
  n = int(np.prod(mask_shape))
  flat = np.ones(n)
  flat[:get_n_zeros(n, sparsity)] = 0
  (random_state.shuffle if random_state else np.random.shuffle)(flat)
  return flat.reshape(mask_shape)


def get_mask_random(mask, sparsity, dtype, random_state=None):
  """sparse_utils.py:71-87.  ``mask`` supplies the shape; returns a NumPy array
  of ``dtype`` (torch dtypes are accepted and mapped)."""
  arr = get_mask_random_numpy(list(mask.shape), sparsity,
                              random_state=random_state)
  return arr.astype(_np_dtype(dtype))


def _np_dtype(dtype):
  try:
    import torch  # pylint: disable=import-outside-toplevel
    if isinstance(dtype, torch.dtype):
      return {torch.float32: np.float32, torch.float64: np.float64,
              torch.int32: np.int32, torch.int64: np.int64,
              torch.bfloat16: np.float32, torch.float16: np.float16}[dtype]
  except ImportError:
    pass
  return dtype


def _erk_raw(shape, include_kernel, power):
  shape = list(shape)
  if include_kernel:
    return (np.sum(shape) / np.prod(shape))**power
  n_in, n_out = shape[-2:]
  return (n_in + n_out) / (n_in * n_out)


def get_sparsities_erdos_renyi(all_masks, default_sparsity, custom_sparsity_map,
                               include_kernel,
                               extract_name_fn=mask_extract_name_fn,
                               erk_power_scale=DEFAULT_ERK_SCALE):
  """Erdos-Renyi(-Kernel) sparsities (sparse_utils.py:90-207).

  Solve  eps * sum_l raw_l * N_l = sum_free (N_l - z_l) - sum_dense z_l  for
  eps, with z_l = floor(s * N_l); whenever eps * max(raw) > 1 every layer at
  the maximum becomes dense and the solve repeats.  sparsity_l = 1 - eps*raw_l.
  """
  info = []
  for m in all_masks:
    var = extract_name_fn(m.name)
    n = np.prod(list(m.shape))
    info.append((m.name, var, n, get_n_zeros(n, default_sparsity),
                 var in custom_sparsity_map))
  dense = set()
  while True:
    rhs = 0
    divisor = 0
    raw = {}
    for (mname, var, n, z, is_custom), m in zip(info, all_masks):
      if var in dense:
        rhs -= z
      elif not is_custom:
        rhs += n - z
        raw[mname] = _erk_raw(m.shape, include_kernel, erk_power_scale)
        divisor += raw[mname] * n
    eps = rhs / divisor
    top = np.max(list(raw.values()))
    if top * eps > 1:
      dense.update(extract_name_fn(k) for k, v in raw.items() if v == top)
    else:
      break
  out = {}
  for mname, var, _, _, is_custom in info:
    if is_custom:
      out[mname] = custom_sparsity_map[var]
    elif var in dense:
      out[mname] = 0.
    else:
      out[mname] = 1. - eps * raw[mname]
  return out


def get_sparsities_uniform(all_masks, default_sparsity, custom_sparsity_map,
                           extract_name_fn=mask_extract_name_fn):
  """sparse_utils.py:210-235."""
  return {m.name: custom_sparsity_map.get(extract_name_fn(m.name),
                                          default_sparsity)
          for m in all_masks}


def get_sparsities(all_masks, method, default_sparsity, custom_sparsity_map,
                   extract_name_fn=mask_extract_name_fn,
                   erk_power_scale=DEFAULT_ERK_SCALE):
  """sparse_utils.py:258-316: 'random' | 'erdos_renyi' | 'erdos_renyi_kernel'
  ('str' -- the hard-coded STR table -- is out of scope)."""
  names = {extract_name_fn(m.name) for m in all_masks}
  missing = set(custom_sparsity_map.keys()) - names
  if missing:
    raise ValueError('No masks are found for the following names: %s' %
                     str(missing))
  if method in ('erdos_renyi', 'erdos_renyi_kernel'):
    return get_sparsities_erdos_renyi(
        all_masks, default_sparsity, custom_sparsity_map,
        include_kernel=(method == 'erdos_renyi_kernel'),
        extract_name_fn=extract_name_fn, erk_power_scale=erk_power_scale)
  if method == 'random':
    return get_sparsities_uniform(all_masks, default_sparsity,
                                  custom_sparsity_map,
                                  extract_name_fn=extract_name_fn)
  raise ValueError('Method: %s is not valid mask initialization method' %
                   method)


def get_mask_init_fn(all_masks, method, default_sparsity, custom_sparsity_map,
                     mask_fn=get_mask_random, erk_power_scale=DEFAULT_ERK_SCALE,
                     extract_name_fn=mask_extract_name_fn):
  """sparse_utils.py:319-364.  Returns a callable that assigns every mask (in
  list order, one RNG shuffle each) and returns the sparsity dict."""
  sparsities = get_sparsities(all_masks, method, default_sparsity,
                              custom_sparsity_map,
                              erk_power_scale=erk_power_scale,
                              extract_name_fn=extract_name_fn)

  def init_fn(*unused_args, **unused_kwargs):
    for m in all_masks:
      m.assign(mask_fn(m, sparsities[m.name], m.dtype))
    return sparsities

  init_fn.sparsities = sparsities
  return init_fn


class StatLayer:
  """What `get_stats` needs of a layer -- the fields it reads off a tf.keras layer (sparse_utils.py:367-447): the
  kernel's name and shape, the layer type, the input's spatial size and the strides."""

  class _Kernel:

    def __init__(self, name, shape):
      self.name = name
      self.shape = tuple(int(s) for s in shape)

  def __init__(self, kind, name, kernel_shape, input_size=None, strides=(1, 1)):
    if kind not in ('conv2d', 'depthwise', 'dense'):
      raise ValueError('kind must be conv2d, depthwise or dense')
    self.kind = kind
    self.kernel = StatLayer._Kernel(name, kernel_shape)
    self.input_shape = (None, input_size, input_size, None)
    self.strides = tuple(strides)


def get_stats(masked_layers, default_sparsity=0.8, method='erdos_renyi', custom_sparsities=None, is_debug=False,
              width=1., first_layer_name='conv1', last_layer_name='conv_preds', param_size=32,
              erk_power_scale=DEFAULT_ERK_SCALE):
  """Size and effective FLOPs of a sparse model (sparse_utils.py:376-454): the per-layer sparsities of `method` at
  `default_sparsity`, then for every layer the MicroNet-challenge count of a sparse dot product (rigl_amd/counting.py).

  ``masked_layers``: `StatLayer`s (or anything with .kernel.name / .kernel.shape / .kind / .input_shape / .strides).
  Returns (total_flops = multiplications + additions of one inference, total_param_bits, real_sparsity) like the reference.
  These are EFFECTIVE flops (zeros skipped) -- the reference's 0.42x at ERK 0.8; the MFMA kernels of this package execute
  the dense-equivalent count, and bench.py reports the two side by side, never mixed."""
  from rigl_amd import counting  # pylint: disable=import-outside-toplevel
  if custom_sparsities is None:
    custom_sparsities = {}
  sparsities = get_sparsities([l.kernel for l in masked_layers], method, default_sparsity, custom_sparsities,
                              lambda a: a, erk_power_scale=erk_power_scale)
  total_flops = 0
  total_param_bits = 0
  total_params = 0.
  n_zeros = 0.
  for layer in masked_layers:
    kernel = layer.kernel
    k_shape = list(kernel.shape)
    d_in, d_out = (0, 1) if len(k_shape) == 2 else (2, 3)
    if not kernel.name.startswith(first_layer_name) and k_shape[d_in] != 1:
      k_shape[d_in] = int(k_shape[d_in] * width)
    if not kernel.name.startswith(last_layer_name) and k_shape[d_out] != 1:
      k_shape[d_out] = int(k_shape[d_out] * width)
    if is_debug:
      print(kernel.name, layer.input_shape, k_shape, sparsities[kernel.name])
    if layer.kind == 'conv2d':
      layer_op = counting.Conv2D(layer.input_shape[1], k_shape, layer.strides, 'same', True, 'relu')
    elif layer.kind == 'depthwise':
      layer_op = counting.DepthWiseConv2D(layer.input_shape[1], k_shape, layer.strides, 'same', True, 'relu')
    elif layer.kind == 'dense':
      layer_op = counting.FullyConnected(k_shape, True, 'relu')
    else:
      raise ValueError('Should not happen.')
    param_count, n_mults, n_adds = counting.count_ops(layer_op, sparsities[kernel.name], param_size)
    total_param_bits += param_count
    total_flops += n_mults + n_adds
    n_param = np.prod(k_shape)
    total_params += n_param
    n_zeros += int(n_param * sparsities[kernel.name])
  return total_flops, total_param_bits, n_zeros / total_params
