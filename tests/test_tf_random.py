"""Stateless-noise parity: Python's salted str hash, Philox-4x32-10
and TensorFlow's stateless_random_{normal,uniform} layout.

Pinned: the hash against the real interpreter (subprocesses under
PYTHONHASHSEED = 0 / 1 / 1234), Philox against the Random123 known-answer
vectors.  Unpinned (TensorFlow not available offline): GenerateKey's scramble and
the sample layout, restated from TF 1.15's sources -- see oracle/tf_random.py."""
import subprocess
import sys

import numpy as np
import pytest

from oracle import tf_random as T

NAMES = ['abc', '', 'resnet_model/initial_conv/weights:0drop', 'resnet_model/final_dense/weights:0grow',
         'x' * 7, 'x' * 8, 'x' * 9, 'x' * 31, 'café', '日本語', '\U0001f600 wide']


@pytest.mark.parametrize('hashseed', [0, 1, 1234])
def test_python_str_hash_matches_the_interpreter(hashseed):
  from rigl_amd import pyhash
  code = 'import sys; print([hash(s) for s in %r])' % (NAMES,)
  out = subprocess.check_output([sys.executable, '-c', code], env={'PYTHONHASHSEED': str(hashseed)})
  want = eval(out.decode())   # noqa: S307 -- our own subprocess printing a list of ints
  assert [T.python_str_hash(s, hashseed) for s in NAMES] == want
  assert [pyhash.python_str_hash(s, hashseed) for s in NAMES] == want


def test_name_hash_follows_an_explicit_pythonhashseed(monkeypatch):
  from rigl_amd import pyhash
  monkeypatch.delenv('PYTHONHASHSEED', raising=False)
  assert pyhash.name_hash('w:0drop') == T.python_str_hash('w:0drop', 0)
  monkeypatch.setenv('PYTHONHASHSEED', 'random')
  assert pyhash.name_hash('w:0drop') == T.python_str_hash('w:0drop', 0)


def test_philox_known_answers():
  """Random123 kat_vectors, philox4x32-10."""
  def run(c, k):
    return ['%08x' % v for v in T.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))]
  assert run([0, 0, 0, 0], [0, 0]) == ['6627e8d5', 'e169c58d', 'bc57ac4c', '9b00dbd8']
  assert run([0xffffffff] * 4, [0xffffffff] * 2) == ['408f276d', '41c83b0e', 'a20bc7c6', '6d5451fd']
  assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
      ['d16cfe09', '94fdcceb', '5001e420', '24126ea1']


def test_stream_layout_and_seed_wrapping():
  a = T.stateless_u32(11, 5, 7)
  b = T.stateless_u32(64, 5, 7)
  assert np.array_equal(a, b[:11])                       # prefix property: element i does not depend on n
  assert not np.array_equal(T.stateless_u32(8, 5, 7), T.stateless_u32(8, 5, 8))
  assert not np.array_equal(T.stateless_u32(8, -1, 7), T.stateless_u32(8, 0xFFFFFFFF >> 1, 7))
  # tf.cast(stack([offset + hash, gs]), int32) wraps
  assert T.tf_seed_pair(3, (1 << 32) + 5, 9) == (8, 9)
  assert T.tf_seed_pair(0, -1, (1 << 31)) == (-1, -(1 << 31))
  s0, _ = T.tf_seed_pair(0, T.python_str_hash('resnet_model/initial_conv/weights:0drop', 0), 0)
  assert -(1 << 31) <= s0 < (1 << 31)


def test_distributions_are_sane():
  n = 200000
  u = T.stateless_random_uniform(n, 11, 3)
  assert u.dtype == np.float32 and 0.0 <= u.min() and u.max() < 1.0
  assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
  z = T.stateless_random_normal(n, 11, 3)
  assert abs(z.mean()) < 1e-2 and abs(z.std() - 1.0) < 1e-2
  assert abs(np.mean(z ** 3)) < 3e-2 and abs(np.mean(z ** 4) - 3.0) < 0.1
  w = T.stateless_random_normal(n, 11, 3, mean=0.0, stddev=1e-5)
  assert np.array_equal(w, (z * np.float32(1e-5)).astype(np.float32))
  v = T.stateless_random_uniform(1000, 2, 2, minval=-0.25, maxval=0.25)
  assert v.min() >= -0.25 and v.max() < 0.25


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 3, 4, 5, 1023, 4096, 100003])
@pytest.mark.parametrize('seed', [(0, 0), (5, 7), (-123456789, 100), (2147483647, -1)])
def test_hip_generator_matches_the_oracle(n, seed):
  import torch
  from rigl_amd import ops
  dev = 'cuda:0'
  u = ops.stateless_random(n, seed[0], seed[1], 'uniform', device=dev).cpu().numpy()
  assert np.array_equal(u.view(np.uint32), T.stateless_random_uniform(n, *seed).view(np.uint32))
  u2 = ops.stateless_random(n, seed[0], seed[1], 'uniform', scale=0.5, shift=-0.25, device=dev).cpu().numpy()
  assert np.array_equal(u2.view(np.uint32), T.stateless_random_uniform(n, seed[0], seed[1], -0.25, 0.25).view(np.uint32))
  z = ops.stateless_random(n, seed[0], seed[1], 'normal', device=dev).cpu().numpy()
  zr = T.stateless_random_normal(n, *seed)
  # sinf / cosf / logf: ocml vs glibc, a couple of ulp on values of magnitude <= ~5.5
  np.testing.assert_allclose(z, zr, rtol=0, atol=2e-6)
  zs = ops.stateless_random(n, seed[0], seed[1], 'normal', scale=1e-5, device=dev).cpu().numpy()
  np.testing.assert_allclose(zs, T.stateless_random_normal(n, seed[0], seed[1], 0.0, 1e-5), rtol=0, atol=2e-11)
  assert torch.is_tensor(ops.stateless_random(0, 1, 2, 'normal', device=dev))


@pytest.mark.gpu
def test_rigl_update_draws_the_reference_noise():
  """SparseRigLOptimizer's drop noise is stateless_random_normal(stddev=1e-5,
  seed=int32([offset + hash(name + 'drop'), global_step])) of the layer."""
  import torch
  from rigl_amd import pyhash, sparse_optimizers as SO, train, variables as V, pruning_layers as PL
  g = V.reset_default_graph('cuda:0')
  layer = PL.MaskedDense(g, 'fc1', 64, 32, use_bias=False, sparsity_technique='threshold')
  g.finalize()
  inner = train.GradientDescentOptimizer(0.1, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 100, 10, drop_fraction=0.3, stateless_seed_offset=17, noise_std=1e-5)
  gs = g.get_or_create_global_step()
  gs.value = 40
  opt._global_step = gs
  lv = g.masked_layers()[0]
  noise = opt._drop_noise(lv, 1e-5).cpu().numpy()
  s0, s1 = T.tf_seed_pair(17, pyhash.name_hash(lv.weights.name + 'drop'), 40)
  np.testing.assert_allclose(noise, T.stateless_random_normal(64 * 32, s0, s1, 0.0, 1e-5), rtol=0, atol=2e-11)
  assert lv.weights.name.endswith(':0')


@pytest.mark.gpu
def test_batched_fill_equals_the_single_calls():
  """rigl_stateless_random_batched: every tensor of a mask update in one launch, bit for bit the per-tensor streams
  (more items than one launch's table holds, empty and ragged sizes, both distributions)."""
  import torch
  from rigl_amd import ops
  dev = 'cuda:0'
  rs = np.random.RandomState(5)
  items = []
  for i in range(70):
    n = int(rs.choice([0, 1, 3, 4, 5, 63, 64, 65, 1000, 4099, 36864, 2359296 if i == 7 else 147]))
    items.append((n, int(rs.randint(-2**31, 2**31)), int(rs.randint(-2**31, 2**31)), 'normal' if i % 3 else 'uniform',
                  float(rs.choice([1.0, 1e-5, 0.5])), float(rs.choice([0.0, -0.25]))))
  outs = ops.stateless_random_batched(items, dev)
  assert len(outs) == len(items)
  for it, o in zip(items, outs):
    assert o.numel() == it[0]
    ref = ops.stateless_random(it[0], it[1], it[2], it[3], scale=it[4], shift=it[5], device=dev)
    assert torch.equal(o, ref), it


@pytest.mark.gpu
def test_whole_model_update_uses_the_batched_noise():
  """mask_update_op prefetches every layer's drop noise in one launch; the tensors are the per-layer streams."""
  import torch
  from rigl_amd import sparse_optimizers as SO, train, variables as V, pruning_layers as PL
  g = V.reset_default_graph('cuda:0')
  for i, (a, b) in enumerate([(64, 32), (32, 48), (48, 8)]):
    PL.MaskedDense(g, 'fc%d' % i, a, b, use_bias=False, sparsity_technique='threshold')
  g.finalize()
  inner = train.GradientDescentOptimizer(0.1, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 100, 10, drop_fraction=0.3, stateless_seed_offset=3, noise_std=1e-5)
  gs = g.get_or_create_global_step()
  gs.value = 20
  opt._global_step = gs
  single = [opt._drop_noise(lv, 1e-5).clone() for lv in g.masked_layers()]
  opt._prefetch_drop_noise(g.masked_layers())
  assert len(opt._noise_cache) == 3
  for lv, s in zip(g.masked_layers(), single):
    assert torch.equal(opt._drop_noise(lv, 1e-5), s)
  opt._noise_cache = {}
