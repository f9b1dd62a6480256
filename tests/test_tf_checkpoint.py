"""TF checkpoint bundle reader / writer.  No TensorFlow and no
released checkpoint are available offline, so parity with real files is
UNPINNED; pinned here: CRC-32C against RFC 3720's vectors, the table footer
magic, reader <-> writer round trips (multi-block tables, dtypes, corruption
detection) and the graph loaders' name handling."""
import os
import struct

import numpy as np
import pytest

from rigl_amd import tf_checkpoint as TC


def test_crc32c_known_answers():
  assert TC.crc32c(b'123456789') == 0xE3069283
  assert TC.crc32c(bytes(32)) == 0x8A9136AA                 # RFC 3720 B.4
  assert TC.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
  assert TC.crc32c(bytes(range(32))) == 0x46DD794E
  assert TC.crc32c(b'6789', TC.crc32c(b'12345')) == 0xE3069283   # incremental
  # leveldb's documented mask: rotate right by 15, add the delta
  c = TC.crc32c(b'foo')
  assert TC.masked_crc32c(b'foo') == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _tensors(rng, n):
  out = {}
  for i in range(n):
    scope = 'resnet_model/bottleneck_%d_block_group%d_%d_1' % (i % 3 + 1, i % 4 + 1, i // 12)
    out[scope + '/weights'] = rng.standard_normal((1, 1, 8 + i % 5, 16)).astype(np.float32)
    out[scope + '/mask'] = (rng.random((1, 1, 8 + i % 5, 16)) < 0.2).astype(np.float32)
  out['global_step'] = np.array(32000, dtype=np.int64)
  out['flags/bools'] = np.array([True, False, True])
  out['small/int32'] = np.arange(-3, 4, dtype=np.int32)
  out['small/f64'] = rng.standard_normal(5)
  out['empty'] = np.zeros((0, 4), np.float32)
  return out


@pytest.mark.parametrize('n,block', [(2, 4096), (40, 256), (120, 4096)])
def test_round_trip(tmp_path, n, block):
  rng = np.random.default_rng(n)
  tensors = _tensors(rng, n)
  prefix = str(tmp_path / 'model.ckpt-32000')
  TC.write_bundle(prefix, tensors, block_bytes=block)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
  r = TC.BundleReader(prefix)
  assert r.keys() == sorted(tensors)
  assert r.get_variable_to_shape_map()['global_step'] == []
  for k, v in tensors.items():
    got = r.get_tensor(k)
    assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
  with pytest.raises(KeyError):
    r.get_tensor('nope')


def test_corruption_is_detected(tmp_path):
  rng = np.random.default_rng(0)
  prefix = str(tmp_path / 'm')
  TC.write_bundle(prefix, _tensors(rng, 6))
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[10] ^= 0x40
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  r = TC.BundleReader(prefix)
  bad = [k for k in r.keys() if _raises(lambda: r.get_tensor(k))]
  assert len(bad) == 1                                     # exactly the tensor that owns byte 10
  idx = bytearray(open(prefix + '.index', 'rb').read())
  idx[5] ^= 0x01
  open(prefix + '.index', 'wb').write(bytes(idx))
  with pytest.raises(TC.CheckpointError):
    TC.BundleReader(prefix)
  open(prefix + '.index', 'wb').write(b'not a table')
  with pytest.raises(TC.CheckpointError):
    TC.BundleReader(prefix)
  with pytest.raises(TC.CheckpointError):
    TC.BundleReader(str(tmp_path / 'missing'))


def _raises(fn):
  try:
    fn()
  except TC.CheckpointError:
    return True
  return False


def test_bfloat16_entries_are_widened(tmp_path):
  """A DT_BFLOAT16 entry (dtype enum 14) written by hand."""
  prefix = str(tmp_path / 'b')
  vals = np.array([1.0, -2.5, 3.140625], np.float32)
  raw = (vals.view(np.uint32) >> 16).astype('<u2').tobytes()
  open(prefix + '.data-00000-of-00001', 'wb').write(raw)
  entry = TC._entry_proto(14, (3,), 0, len(raw), TC.masked_crc32c(raw))
  header = TC._field(1, 0, TC._put_varint(1))
  block = TC._build_block([(b'', header), (b'x', entry)])
  out = bytearray()
  h_data = TC._put_varint(0) + TC._put_varint(len(block))
  out += block + b'\x00' + struct.pack('<I', TC.masked_crc32c(block + b'\x00'))
  meta = TC._build_block([])
  h_meta = TC._put_varint(len(out)) + TC._put_varint(len(meta))
  out += meta + b'\x00' + struct.pack('<I', TC.masked_crc32c(meta + b'\x00'))
  index = TC._build_block([(b'x', h_data)], restart_interval=1)
  h_index = TC._put_varint(len(out)) + TC._put_varint(len(index))
  out += index + b'\x00' + struct.pack('<I', TC.masked_crc32c(index + b'\x00'))
  footer = h_meta + h_index
  out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TC.TABLE_MAGIC)
  open(prefix + '.index', 'wb').write(bytes(out))
  got = TC.BundleReader(prefix).get_tensor('x')
  assert got.dtype == np.float32 and np.array_equal(got, vals)


@pytest.mark.gpu
def test_graph_save_load_and_partial_mask_loading(tmp_path):
  """save_graph -> a fresh graph: everything; then only '*mask' into a third one
  (initialize_parameters_from_ckpt, imagenet_resnet/utils.py:93-125)."""
  torch = pytest.importorskip('torch')
  from rigl_amd import pruning_layers as PL, sparse_utils, variables as V
  from rigl_amd.workloads import nn as gnn

  def build(seed):
    g = V.Graph('cuda:0')
    PL.set_init_seed(seed)
    PL.MaskedDense(g, 'resnet_model/fc_a', 24, 16, use_bias=True, sparsity_technique='threshold')
    gnn.BatchNorm(g, 'resnet_model/bn_a', 16)
    PL.MaskedDense(g, 'resnet_model/fc_b', 16, 8, use_bias=False, sparsity_technique='threshold')
    gnn.BatchNorm(g, 'resnet_model/bn_b', 8)
    g.finalize()
    return g

  src = build(1)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(src.get_masks(), 'random', 0.5, {})()
  prefix = str(tmp_path / 'ckpt')
  names = TC.save_graph(prefix, src)
  assert 'resnet_model/fc_a/mask' in names and 'resnet_model/bn_a/moving_variance' in names
  dst = build(2)
  loaded = TC.load_into_graph(prefix, dst, strict=True)
  assert set(loaded) == set(names)
  for a, b in zip(src.masked_layers(), dst.masked_layers()):
    assert np.array_equal(a.mask.numpy(), b.mask.numpy())
    assert torch.equal(a.weights.data, b.weights.data)
  only = build(3)
  w_before = [l.weights.data.clone() for l in only.masked_layers()]
  got = TC.initialize_parameters_from_ckpt(prefix, only, 'mask')
  assert sorted(got) == ['resnet_model/fc_a/mask', 'resnet_model/fc_b/mask']
  for l, a, w0 in zip(only.masked_layers(), src.masked_layers(), w_before):
    assert np.array_equal(l.mask.numpy(), a.mask.numpy()) and torch.equal(l.weights.data, w0)
  nm = TC.tf1_batch_norm_name_map(src)
  assert nm['resnet_model/bn_a/gamma'] == 'resnet_model/batch_normalization/gamma'
  assert nm['resnet_model/bn_b/moving_mean'] == 'resnet_model/batch_normalization_1/moving_mean'
