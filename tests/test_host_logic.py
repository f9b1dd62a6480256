"""CPU tests of the product's host logic (no device compute): sparsity
distributions and mask init vs reference-generated goldens, the optimizer
schedule vs the reference's golden sequences, API errors, workload tables."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

from rigl_amd import sparse_utils as SU

G = os.path.join(os.path.dirname(__file__), 'golden')


class FakeMask:

  def __init__(self, name, shape):
    self.name, self.shape, self.dtype = name, tuple(shape), np.float32
    self.value = None

  def assign(self, v):
    self.value = np.asarray(v)


def _unpack(hexbits, n):
  b = np.frombuffer(bytes.fromhex(hexbits), dtype=np.uint8)
  return np.unpackbits(b, bitorder='little')[:n]


def test_get_sparsities_bitwise_vs_reference():
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  for r in data['runs']:
    masks = [FakeMask(n, s) for n, s in zip(r['names'], r['shapes'])]
    got = SU.get_sparsities(masks, r['method'], r['default_sparsity'],
                            r['custom'], erk_power_scale=r['erk_power_scale'])
    for name, hx in zip(r['names'], r['sparsities']):
      assert float(got[name]) == float.fromhex(hx), (r['net'], r['method'], name)


def test_get_sparsities_errors():
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  masks = [FakeMask('var1/mask:0', (2, 4)), FakeMask('var2/mask:0', (2, 3)),
           FakeMask('var3/mask:0', (1, 1, 3))]
  for e in data['errors']:
    with pytest.raises(ValueError) as ei:
      SU.get_sparsities(masks, e['method'], 0.5, e['custom'])
    assert str(ei.value) == e['error']


def test_get_mask_random_bits_vs_reference():
  for c in json.load(open(os.path.join(G, 'mask_random.json'))):
    n = int(np.prod(c['shape']))
    if c['seed'] == 'global11':
      np.random.seed(11)
      m = SU.get_mask_random(FakeMask('a/mask:0', c['shape']), c['sparsity'], np.int32)
      assert m.dtype == np.int32
    else:
      m = SU.get_mask_random_numpy(c['shape'], c['sparsity'], np.random.RandomState(c['seed']))
    assert int(m.sum()) == c['ones']
    np.testing.assert_array_equal(m.reshape(-1).astype(np.uint8), _unpack(c['bits'], n))


@pytest.mark.parametrize('dtype', [np.int32, np.float32, np.int64, np.float64])
def test_mask_dtype(dtype):
  # rigl/sparse_utils_test.py:57-62
  assert SU.get_mask_random(FakeMask('a/mask:0', (3, 2)), 0.5, dtype).dtype == dtype


def test_mask_init_fn_order_and_counts():
  """get_mask_init_fn: masks assigned in list order with one shuffle each from
  the global RNG (sparse_utils.py:359-362) -- same stream as the oracle."""
  from oracle import rigl_oracle as O
  shapes = OrderedDict([('l1/mask:0', (784, 300)), ('l2/mask:0', (300, 100)), ('l3/mask:0', (100, 10))])
  masks = [FakeMask(n, s) for n, s in shapes.items()]
  np.random.seed(5)
  SU.get_mask_init_fn(masks, 'random', 0.9, {'l3': 0.0})()
  np.random.seed(5)
  ref, _ = O.get_mask_init(shapes, 'random', 0.9, {'l3': 0.0})
  for m in masks:
    np.testing.assert_array_equal(m.value, ref[m.name])
  assert masks[2].value.sum() == 1000


# ---------------------------------------------------------------- optimizer schedule (host scalars)
def _opt(begin, end, freq, frac=0.5, anneal='constant', cls='rigl'):
  import torch
  from rigl_amd import sparse_optimizers as SO, train, variables as V
  g = V.reset_default_graph('cpu')
  inner = train.GradientDescentOptimizer(0.1, graph=g)
  c = SO.SparseRigLOptimizer if cls == 'rigl' else SO.SparseSETOptimizer
  return c(inner, begin, end, freq, drop_fraction=frac, drop_fraction_anneal=anneal), g


def test_rigl_schedule_golden_sequences():
  """rigl/sparse_optimizers_test.py:349-367 through the product's own
  is_mask_update_iter (no kernels: the update/apply actions are stubbed)."""
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  for c in sched['rigl_increment']:
    opt, g = _opt(c['begin'], c['end'], c['freq'])
    gs = g.get_or_create_global_step()
    opt.mask_update_op = lambda step: None
    seq = []
    for _ in c['seq']:
      before = gs.value
      opt._global_step = gs
      opt.cond_mask_update_op(gs, lambda: setattr(gs, 'value', gs.value + 1))
      seq.append(gs.value - before)
    assert seq == c['seq']


def test_set_schedule_golden():
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  for c in sched['set_updates']:
    opt, g = _opt(c['begin'], c['end'], c['freq'], cls='set')
    gs = g.get_or_create_global_step()
    fired = []
    opt.mask_update_op = lambda step: fired.append(step)
    for _ in range(c['iters']):
      gs.value += 1                       # inner optimizer applied first (:133-146)
      opt.cond_mask_update_op(gs, lambda: None)
    assert fired == c['changed']


def test_drop_fraction_bits_vs_reference():
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  for c in sched['drop_fraction']:
    opt, _ = _opt(c['begin'], c['end'], 1, frac=c['init'], anneal=c['anneal'])
    for step, flag, hx in c['values']:
      got = opt.get_drop_fraction(step, flag)
      assert np.float32(got) == np.float32(float.fromhex(hx)), (c['anneal'], step)
  opt, _ = _opt(0, 5, 1, anneal='bogus')
  with pytest.raises(ValueError) as ei:
    opt.get_drop_fraction(0, True)
  assert str(ei.value) == sched['bad_anneal_error']


@pytest.mark.parametrize('method', ['ones', 'zero', None, 0])
def test_grow_tensor_value_error(method):
  # rigl/sparse_optimizers_test.py:181-189
  import torch
  opt, _ = _opt(0, 0, 1, cls='set')
  with pytest.raises(ValueError):
    opt.get_grow_tensor(torch.rand(3, 4), method)


def test_extract_number():
  from rigl_amd.sparse_optimizers import extract_number
  assert extract_number('foo_.5') == 0.5 and extract_number('foo_foo.5') == 1.0
  assert extract_number('foo_0.5') == 0.5 and extract_number('foo_4') == 4
  assert extract_number('grad_scale_2') == 2 and extract_number('zeros') == 1.0


def test_layer_api_errors():
  import torch
  from rigl_amd import pruning_layers as PL, variables as V
  V.reset_default_graph('cpu')
  x = torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16)
  with pytest.raises(ValueError):
    PL.sparse_conv2d(x, 8, [3, 3], sparsity_technique='bogus')
  with pytest.raises(ValueError):
    PL.sparse_conv2d(x, 8, [3, 3], data_format='channels_middle')
  with pytest.raises(ValueError):
    PL.sparse_conv2d(x[0], 8, [3, 3])
  with pytest.raises(ValueError):
    PL.sparse_fully_connected(x, 8, sparsity_technique='bogus')


def test_reference_alias_package():
  import rigl.sparse_optimizers as a
  import rigl.sparse_utils as b
  from rigl.imagenet_resnet import pruning_layers as c
  from rigl_amd import sparse_optimizers, sparse_utils, pruning_layers
  assert a.SparseRigLOptimizer is sparse_optimizers.SparseRigLOptimizer
  assert a.get_grow_grads is sparse_optimizers.get_grow_grads
  assert b.get_mask_random is sparse_utils.get_mask_random
  assert b.get_stats is sparse_utils.get_stats
  assert c.sparse_conv2d is pruning_layers.sparse_conv2d


def test_workload_tables_match_reference_sizes():
  """The product's ResNet-50 definition creates exactly the masks (names,
  HWIO shapes, order) of the reference; sizes pinned by the reference's own
  table in rigl/str_sparsities.py."""
  from rigl_amd.workloads import shapes as WS
  from tests.golden import layer_shapes as LS
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  table = {n: p for n, p, _ in data['resnet50_size_table']}
  r50 = WS.resnet50_masks()
  assert list(r50.items()) == list(LS.resnet50().items())
  assert {n: int(np.prod(s)) for n, s in r50.items()} == table
  assert list(WS.mobilenet_v1_masks().items()) == list(LS.mobilenet_v1().items())
  assert list(WS.wide_resnet_masks(22, 1).items()) == list(LS.wide_resnet(22, 1).items())
  assert sum(int(np.prod(s)) for s in WS.wide_resnet_masks(22, 1).values()) == 270464
  assert sum(int(np.prod(s)) for s in WS.mobilenet_v1_masks().values()) == 4163584


def test_glue_compositions_fall_back_to_the_plain_ops_off_the_gpu():
  """nn.bn_add_bn_relu / nn.bn_relu_max_pool_3x3_s2_same are single nodes on the GPU (bn.hip / pool.hip); anywhere else --
  fp32 tensors on the host, evaluation mode -- they must be exactly the composition they replace
  (resnet_model.py:456-501, :631-644)."""
  import torch
  from rigl_amd import variables as V
  from rigl_amd.workloads import nn as gnn
  g = V.Graph('cpu')
  bn, bn2 = gnn.BatchNorm(g, 'a', 16), gnn.BatchNorm(g, 'b', 16)
  g.finalize()
  gen = torch.Generator().manual_seed(0)
  for b in (bn, bn2):
    b.gamma.data.copy_(torch.rand(16, generator=gen) + 0.5)
    b.beta.data.copy_(torch.randn(16, generator=gen) * 0.1)
  x, x2 = torch.randn(2, 6, 6, 16, generator=gen), torch.randn(2, 6, 6, 16, generator=gen)
  for training in (True, False):
    # (training-mode outputs use batch statistics, evaluation mode does not touch the moving ones: calling twice is fair)
    got = gnn.bn_add_bn_relu(bn, x, bn2, x2, training)
    want = bn(x, training, relu=True, residual=bn2(x2, training, relu=False))
    assert torch.equal(got, want)
    got = gnn.bn_relu_max_pool_3x3_s2_same(bn, x, training)
    want = gnn.max_pool_3x3_s2_same(bn(x, training, relu=True))
    assert got.shape == (2, 3, 3, 16) and torch.equal(got, want)


def test_precision_flag_maps_to_the_activation_dtype():
  """--precision of the reference (imagenet_train_eval.py:56-59): bfloat16 | float32, anything else is an error."""
  import torch
  from rigl_amd.workloads import nn as gnn
  assert gnn.activation_dtype('bfloat16') == torch.bfloat16
  assert gnn.activation_dtype('float32') == torch.float32
  with pytest.raises(ValueError):
    gnn.activation_dtype('float16')
