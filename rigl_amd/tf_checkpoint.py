"""Reading (and writing) TensorFlow checkpoint bundles without TensorFlow, to
bring the reference's released TF1 checkpoints (README.md:34-58, 81-93, 155;
variables ``resnet_model/<layer>/{weights,mask}``) into this framework's
arenas -- SURVEY 8(f).3.  Mirrors the partial loaders of
rigl/imagenet_resnet/utils.py:93-125 (``initialize_parameters_from_ckpt``).

Format ("tensor bundle", tensorflow/core/util/tensor_bundle, TF >= 0.12):
  <prefix>.index                   a LevelDB-format immutable table (SSTable):
                                   key ""   -> BundleHeaderProto
                                   key name -> BundleEntryProto {dtype, shape,
                                               shard_id, offset, size, crc32c}
  <prefix>.data-SSSSS-of-NNNNN     the raw little-endian tensor bytes
Table layout: data blocks of prefix-compressed entries (varint32 shared /
non_shared / value_len, restart array, 1-byte compression type + masked CRC-32C
trailer), an index block, a metaindex block and a 48-byte footer ending in the
magic 0xdb4775248b80fb57.  Checksums are CRC-32C, "masked"
(rot-right 15 + 0xa282ead8); every block and every tensor is verified on read.

PARITY UNPINNED: no checkpoint file ships with /root/reference and there is no
network, so the reader is validated only against this module's own writer
(which follows the same published format) and the format's fixed points that
can be checked offline: the CRC-32C test vectors of RFC 3720 and the footer
magic.  Snappy-compressed blocks (not produced by TF's BundleWriter) are
rejected with a clear error.
"""
import ctypes as C
import os
import struct

import numpy as np

from rigl_amd import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16,
           6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16,
           22: np.uint32, 23: np.uint64}
_DT_BFLOAT16 = 14
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
  pass


def crc32c(data, crc=0):
  buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
  return int(_lib.load().rigl_crc32c(C.c_char_p(bytes(buf)), len(buf), crc))


def masked_crc32c(data):
  c = crc32c(data)
  return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- varints / minimal protobuf ------------------------------------------------------
def _get_varint(buf, pos):
  out, shift = 0, 0
  while True:
    if pos >= len(buf):
      raise CheckpointError('truncated varint')
    b = buf[pos]
    pos += 1
    out |= (b & 0x7F) << shift
    if not b & 0x80:
      return out, pos
    shift += 7
    if shift > 63:
      raise CheckpointError('varint too long')


def _put_varint(v):
  v &= (1 << 64) - 1
  out = bytearray()
  while v >= 0x80:
    out.append((v & 0x7F) | 0x80)
    v >>= 7
  out.append(v)
  return bytes(out)


def _parse_proto(buf):
  """-> {field: [values]}; varint -> int, 64/32-bit -> int, length-delimited -> bytes."""
  out, pos = {}, 0
  while pos < len(buf):
    tag, pos = _get_varint(buf, pos)
    field, wire = tag >> 3, tag & 7
    if wire == 0:
      v, pos = _get_varint(buf, pos)
    elif wire == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + n])
      if len(v) != n:
        raise CheckpointError('truncated protobuf field')
      pos += n
    else:
      raise CheckpointError('unsupported protobuf wire type %d' % wire)
    out.setdefault(field, []).append(v)
  return out


def _signed64(v):
  return v - (1 << 64) if v >> 63 else v


def _parse_shape(buf):
  dims = []
  for d in _parse_proto(buf).get(2, []):
    dims.append(_signed64(_parse_proto(d).get(1, [0])[0]))
  return tuple(dims)


def _field(tag, wire, payload):
  return _put_varint((tag << 3) | wire) + payload


def _entry_proto(dtype, shape, offset, size, crc):
  shp = b''.join(_field(2, 2, _put_varint(len(d)) + d)
                 for d in (_field(1, 0, _put_varint(int(s))) for s in shape))
  out = _field(1, 0, _put_varint(dtype)) + _field(2, 2, _put_varint(len(shp)) + shp)
  if offset:
    out += _field(4, 0, _put_varint(offset))
  out += _field(5, 0, _put_varint(size)) + _field(6, 5, struct.pack('<I', crc))
  return out


# ---- table (SSTable) -------------------------------------------------------------------
def _read_block(buf, offset, size, what):
  raw = buf[offset:offset + size + 5]
  if len(raw) != size + 5:
    raise CheckpointError('%s: block [%d, +%d) beyond the end of the index file' % (what, offset, size))
  body, ctype, crc = raw[:size], raw[size], struct.unpack('<I', raw[size + 1:size + 5])[0]
  if masked_crc32c(raw[:size + 1]) != crc:
    raise CheckpointError('%s: block checksum mismatch' % what)
  if ctype != 0:
    raise CheckpointError('%s: compressed table blocks (type %d) are not supported' % (what, ctype))
  return body


def _block_entries(block):
  if len(block) < 4:
    raise CheckpointError('table block too small')
  n_restarts = struct.unpack('<I', block[-4:])[0]
  limit = len(block) - 4 - 4 * n_restarts
  if limit < 0:
    raise CheckpointError('bad restart array')
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise CheckpointError('corrupt table entry')
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    yield key, bytes(block[pos:pos + vlen])
    pos += vlen


def _table_items(buf):
  if len(buf) < 48:
    raise CheckpointError('index file shorter than a table footer')
  footer = buf[-48:]
  if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
    raise CheckpointError('not a TensorFlow checkpoint index (bad table magic)')
  _, p = _get_varint(footer, 0)          # metaindex handle (unused)
  _, p = _get_varint(footer, p)
  ioff, p = _get_varint(footer, p)
  isize, p = _get_varint(footer, p)
  for _, handle in _block_entries(_read_block(buf, ioff, isize, 'index block')):
    off, q = _get_varint(handle, 0)
    size, _ = _get_varint(handle, q)
    for kv in _block_entries(_read_block(buf, off, size, 'data block')):
      yield kv


class BundleReader:
  """tf.train.NewCheckpointReader twin for a checkpoint prefix
  (``.../model.ckpt-32000``)."""

  def __init__(self, prefix):
    self.prefix = prefix
    try:
      with open(prefix + '.index', 'rb') as fh:
        buf = fh.read()
    except OSError as e:
      raise CheckpointError('cannot open %s.index: %s' % (prefix, e))
    self._entries = {}
    self.num_shards = 1
    for key, value in _table_items(buf):
      msg = _parse_proto(value)
      if key == b'':
        self.num_shards = msg.get(1, [1])[0]
        if msg.get(2, [0])[0] != 0:
          raise CheckpointError('big-endian bundles are not supported')
        continue
      if 7 in msg:
        raise CheckpointError('%s: sliced (partitioned) variables are not supported' % key.decode())
      self._entries[key.decode('utf-8')] = dict(
          dtype=msg.get(1, [0])[0], shape=_parse_shape(msg[2][0]) if 2 in msg else (),
          shard=msg.get(3, [0])[0], offset=msg.get(4, [0])[0], size=msg.get(5, [0])[0],
          crc=msg.get(6, [0])[0])

  def keys(self):
    return sorted(self._entries)

  def has_tensor(self, name):
    return name in self._entries

  def get_variable_to_shape_map(self):
    return {k: list(v['shape']) for k, v in self._entries.items()}

  def get_tensor(self, name, verify=True):
    e = self._entries.get(name)
    if e is None:
      raise KeyError(name)
    path = '%s.data-%05d-of-%05d' % (self.prefix, e['shard'], self.num_shards)
    with open(path, 'rb') as fh:
      fh.seek(e['offset'])
      raw = fh.read(e['size'])
    if len(raw) != e['size']:
      raise CheckpointError('%s: data file truncated' % name)
    if verify and masked_crc32c(raw) != e['crc']:
      raise CheckpointError('%s: tensor checksum mismatch' % name)
    if e['dtype'] == _DT_BFLOAT16:
      arr = (np.frombuffer(raw, dtype='<u2').astype(np.uint32) << 16).view(np.float32)
    elif e['dtype'] in _DTYPES:
      arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e['dtype']]).newbyteorder('<'))
    else:
      raise CheckpointError('%s: unsupported dtype enum %d' % (name, e['dtype']))
    if int(np.prod(e['shape'], dtype=np.int64)) != arr.size:
      raise CheckpointError('%s: shape %s does not match %d stored elements' % (name, e['shape'], arr.size))
    return arr.reshape(e['shape']).copy()


# ---- writer ------------------------------------------------------------------------------
def _build_block(items, restart_interval=16):
  out, restarts, prev = bytearray(), [], b''
  for i, (key, value) in enumerate(items):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
        shared += 1
    out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
    out += key[shared:] + value
    prev = key
  if not restarts:
    restarts = [0]
  out += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
  return bytes(out)


def write_bundle(prefix, tensors, block_bytes=4096):
  """Writes {name: ndarray} as a single-shard bundle that BundleReader -- and
  TensorFlow's tf.train.load_checkpoint -- can open."""
  names = sorted(tensors, key=lambda s: s.encode('utf-8'))
  entries, offset = [], 0
  os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
  with open(prefix + '.data-00000-of-00001', 'wb') as fh:
    for name in names:
      arr = np.asarray(tensors[name], order='C')
      dt = _DT_OF.get(arr.dtype)
      if dt is None:
        raise CheckpointError('%s: dtype %s cannot be stored' % (name, arr.dtype))
      raw = arr.astype(arr.dtype.newbyteorder('<'), copy=False).tobytes()
      fh.write(raw)
      entries.append((name.encode('utf-8'), _entry_proto(dt, arr.shape, offset, len(raw), masked_crc32c(raw))))
      offset += len(raw)
  header = _field(1, 0, _put_varint(1)) + _field(3, 2, (lambda v: _put_varint(len(v)) + v)(_field(1, 0, _put_varint(1))))
  items = [(b'', header)] + entries
  out = bytearray()
  index_items = []

  def emit(block):
    handle = _put_varint(len(out)) + _put_varint(len(block))
    out.extend(block + b'\x00' + struct.pack('<I', masked_crc32c(block + b'\x00')))
    return handle

  cur, cur_size = [], 0
  for kv in items:
    cur.append(kv)
    cur_size += len(kv[0]) + len(kv[1]) + 3
    if cur_size >= block_bytes:
      index_items.append((cur[-1][0], emit(_build_block(cur))))
      cur, cur_size = [], 0
  if cur:
    index_items.append((cur[-1][0], emit(_build_block(cur))))
  meta = emit(_build_block([]))
  index = emit(_build_block(index_items, restart_interval=1))
  footer = meta + index
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out.extend(footer)
  with open(prefix + '.index', 'wb') as fh:
    fh.write(bytes(out))


# ---- graph <-> checkpoint ----------------------------------------------------------------
def variable_map(graph):
  """{TF variable name (no ':0'): object} for everything a TF1 RigL checkpoint of
  this graph would hold: kernels, masks, BN parameters / moving statistics, biases."""
  out = {}
  for v in graph.variables.values():
    out[v.name.split(':')[0]] = v
  for l in graph.masked_layers():
    out[l.mask.name.split(':')[0]] = l.mask
  for scope, mod in graph.modules.items():
    if hasattr(mod, 'moving_mean') and hasattr(mod, 'moving_variance'):
      out[scope + '/moving_mean'] = _Buffer(mod.moving_mean)
      out[scope + '/moving_variance'] = _Buffer(mod.moving_variance)
  return out


class _Buffer:
  """A plain tensor (BN moving statistics) behind the Variable surface used here."""

  def __init__(self, t):
    self.data, self.shape = t, tuple(t.shape)


def tf1_batch_norm_name_map(graph, prefix='resnet_model'):
  """{this graph's BN variable name: the reference checkpoint's name}.  The
  reference creates its BNs with tf.layers.batch_normalization under one
  variable scope, so TF numbers them in creation order:
  ``<prefix>/batch_normalization[_k]/{gamma,beta,moving_mean,moving_variance}``
  (resnet_model.py:41-82); this framework's BatchNorm modules are created in
  the same order (shortcut BN before bn1..3, like bottleneck_block_)."""
  out, k = {}, 0
  for scope, mod in graph.modules.items():
    if not (hasattr(mod, 'moving_mean') and hasattr(mod, 'gamma')):
      continue
    tf_scope = '%s/batch_normalization%s' % (prefix, '_%d' % k if k else '')
    for leaf in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
      out['%s/%s' % (scope, leaf)] = '%s/%s' % (tf_scope, leaf)
    k += 1
  return out


def load_into_graph(prefix, graph, param_suffixes=None, name_map=None, strict=False):
  """Assigns every checkpoint tensor whose (mapped) name matches a graph variable
  or mask.  ``param_suffixes``: str or tuple, restrict to names ending with it
  (e.g. 'mask' to take only the topology, utils.py:93-125).  ``name_map``:
  optional {graph name: checkpoint name}.  Returns the list of loaded names."""
  reader = BundleReader(prefix)
  graph.finalize()
  loaded = []
  suffixes = (param_suffixes,) if isinstance(param_suffixes, str) else param_suffixes
  for gname, obj in sorted(variable_map(graph).items()):
    if suffixes and not gname.endswith(tuple(suffixes)):
      continue
    cname = (name_map or {}).get(gname, gname)
    if not reader.has_tensor(cname):
      if strict:
        raise CheckpointError('checkpoint has no tensor %r' % cname)
      continue
    arr = reader.get_tensor(cname)
    if tuple(arr.shape) != tuple(obj.shape):
      raise CheckpointError('%s: checkpoint shape %s != variable shape %s' % (cname, arr.shape, tuple(obj.shape)))
    if hasattr(obj, 'bits'):
      obj.assign(arr.astype(np.float32))
    else:
      import torch  # pylint: disable=import-outside-toplevel
      with torch.no_grad():
        obj.data.copy_(torch.from_numpy(arr.astype(np.float32)).to(obj.data.device))
    loaded.append(gname)
  graph.shadows_dirty = True
  return loaded


def initialize_parameters_from_ckpt(ckpt_path, graph, param_suffixes):
  """utils.initialize_parameters_from_ckpt twin (imagenet_resnet/utils.py:93-125):
  loads the variables whose names end with ``param_suffixes`` (e.g. 'mask')."""
  return load_into_graph(ckpt_path, graph, param_suffixes)


def save_graph(prefix, graph, extra=None):
  """Writes kernels, masks (as fp32 0/1, the reference's representation) and the
  other variables of ``graph`` under their TF names."""
  graph.finalize()
  tensors = {}
  for name, obj in variable_map(graph).items():
    if hasattr(obj, 'bits'):
      tensors[name] = obj.numpy().astype(np.float32)
    else:
      tensors[name] = obj.data.detach().cpu().numpy().astype(np.float32)
  tensors.update(extra or {})
  write_bundle(prefix, tensors)
  return sorted(tensors)
