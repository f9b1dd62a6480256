"""TEST INFRASTRUCTURE (oracle): NumPy restatement of the random sources RigL's
mask update draws from, so that a fixed-PYTHONHASHSEED reference run can be
matched bit for bit.

The reference (rigl/sparse_optimizers_base.py:402-418, 260-274, 523-534) calls

    stateless_random_normal(shape, stddev=noise_std,
                            seed=int32([offset + hash(weights.name + 'drop'), global_step]))
    stateless_random_uniform(shape, seed=int32([offset + hash(weights.name + 'grow'), global_step]))

Three pieces, each restated here:

 1. Python's ``hash(str)``: SipHash-2-4 over the string's UCS1/2/4 buffer keyed
    by PYTHONHASHSEED (CPython Python/pyhash.c, Python/bootstrap_hash.c
    ``lcg_urandom``).  PINNED in tests against the real interpreter
    (subprocesses with PYTHONHASHSEED=0/1/1234).
 2. Philox-4x32-10 (Salmon et al., SC'11).  PINNED against the Random123
    known-answer vectors.
 3. TensorFlow's use of it -- TensorFlow is a third-party dependency that is
    NOT vendored in /root/reference and not installable here; restated from
    tensorflow/core/kernels/stateless_random_ops.cc (GenerateKey),
    tensorflow/core/lib/random/philox_random.h, random_distributions.h
    (Uint32ToFloat, BoxMullerFloat, 4 samples per counter value, element i
    <- counter + i/4, lane i%4), as of TF 1.15 (the reference's pinned
    tensorflow-gpu==1.15, /root/reference/requirements.txt).
    PARITY UNPINNED for this layer: no TF golden vectors are available offline.
    sinf/cosf/logf are libm's; TF's CPU kernel uses the same glibc calls, the
    HIP kernel uses ocml (<= 2 ulp apart; tests compare with that tolerance).
"""
import numpy as np

U32 = np.uint32
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


# ---------------------------------------------------------------- 1. hash(str)
def _rotl64(x, b):
  return ((x << b) | (x >> (64 - b))) & M64


def siphash24(data, k0, k1):
  """SipHash-2-4 of ``data`` (bytes) -> unsigned 64-bit (pyhash.c:siphash24)."""
  v0 = k0 ^ 0x736f6d6570736575
  v1 = k1 ^ 0x646f72616e646f6d
  v2 = k0 ^ 0x6c7967656e657261
  v3 = k1 ^ 0x7465646279746573

  def rounds(v0, v1, v2, v3, n):
    for _ in range(n):
      v0 = (v0 + v1) & M64; v1 = _rotl64(v1, 13); v1 ^= v0; v0 = _rotl64(v0, 32)
      v2 = (v2 + v3) & M64; v3 = _rotl64(v3, 16); v3 ^= v2
      v0 = (v0 + v3) & M64; v3 = _rotl64(v3, 21); v3 ^= v0
      v2 = (v2 + v1) & M64; v1 = _rotl64(v1, 17); v1 ^= v2; v2 = _rotl64(v2, 32)
    return v0, v1, v2, v3

  n = len(data)
  b = (n & 0xFF) << 56
  full = n - (n % 8)
  for i in range(0, full, 8):
    m = int.from_bytes(data[i:i + 8], 'little')
    v3 ^= m
    v0, v1, v2, v3 = rounds(v0, v1, v2, v3, 2)
    v0 ^= m
  b |= int.from_bytes(data[full:], 'little')
  v3 ^= b
  v0, v1, v2, v3 = rounds(v0, v1, v2, v3, 2)
  v0 ^= b
  v2 ^= 0xff
  v0, v1, v2, v3 = rounds(v0, v1, v2, v3, 4)
  return (v0 ^ v1 ^ v2 ^ v3) & M64


def _hash_secret(hashseed):
  """First 16 bytes of _Py_HashSecret for PYTHONHASHSEED=hashseed
  (bootstrap_hash.c: all zeros for 0, else lcg_urandom)."""
  if hashseed == 0:
    return 0, 0
  x = hashseed & M32
  out = bytearray()
  for _ in range(24):
    x = (x * 214013 + 2531011) & M32
    out.append((x >> 16) & 0xFF)
  return int.from_bytes(out[0:8], 'little'), int.from_bytes(out[8:16], 'little')


def python_str_hash(text, hashseed=0):
  """``hash(text)`` of a CPython >= 3.4, < 3.11 process started with
  PYTHONHASHSEED=hashseed (signed 64-bit)."""
  if not text:
    return 0
  top = max(ord(c) for c in text)
  if top < 256:
    data = text.encode('latin-1')
  elif top < 65536:
    data = text.encode('utf-16-le')
  else:
    data = text.encode('utf-32-le')
  k0, k1 = _hash_secret(hashseed)
  h = siphash24(data, k0, k1)
  if h >= 1 << 63:
    h -= 1 << 64
  return -2 if h == -1 else h


def tf_seed_pair(seed_offset, name_hash, global_step):
  """int32 cast of stack([offset + hash, global_step]) (sparse_optimizers_base.py:404-407)."""
  def i32(v):
    v &= M32
    return v - (1 << 32) if v >= 1 << 31 else v
  return i32(int(seed_offset) + int(name_hash)), i32(int(global_step))


# ---------------------------------------------------------------- 2. Philox-4x32-10
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
  """counter: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
  c = np.array(counter, dtype=np.uint64, copy=True)
  k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint64), c.shape[:-1] + (2,)), copy=True)
  for r in range(10):
    if r:
      k[..., 0] = (k[..., 0] + PHILOX_W0) & M32
      k[..., 1] = (k[..., 1] + PHILOX_W1) & M32
    p0 = PHILOX_M0 * c[..., 0]
    p1 = PHILOX_M1 * c[..., 2]
    n0 = (p1 >> np.uint64(32)) ^ c[..., 1] ^ k[..., 0]
    n1 = p1 & M32
    n2 = (p0 >> np.uint64(32)) ^ c[..., 3] ^ k[..., 1]
    n3 = p0 & M32
    c[..., 0], c[..., 1], c[..., 2], c[..., 3] = n0, n1, n2, n3
  return c.astype(np.uint32)


# ---------------------------------------------------------------- 3. TF stateless ops
def tf_generate_key(seed0, seed1):
  """stateless_random_ops.cc GenerateKey: (key[2], counter[4]) from an int32 seed pair."""
  s0, s1 = int(seed0) & M64, int(seed1) & M64          # int32 -> uint64 sign-extends
  cnt = np.array([s0 & M32, s0 >> 32, s1 & M32, s1 >> 32], dtype=np.uint32)
  mix = philox4x32_10(cnt, np.array([0x3ec8f720, 0x02461e29], dtype=np.uint32))
  key = np.array([mix[0], mix[1]], dtype=np.uint32)
  counter = np.array([0, 0, mix[2], mix[3]], dtype=np.uint32)
  return key, counter


def stateless_u32(n, seed0, seed1):
  """The first n uint32 of the stream: element i = lane i%4 of Philox(counter + i/4)."""
  key, counter = tf_generate_key(seed0, seed1)
  groups = (int(n) + 3) // 4
  g = np.arange(groups, dtype=np.uint64)
  base = (int(counter[0]) | (int(counter[1]) << 32))
  lo = (base + g) & np.uint64(M64)
  # 128-bit counter: carry from the low 64 bits into the high 64 (never reached for real tensor sizes,
  # base's low half is 0 after GenerateKey; kept for completeness)
  carry = (lo < g).astype(np.uint64)
  hi = (np.uint64(int(counter[2]) | (int(counter[3]) << 32)) + carry) & np.uint64(M64)
  c = np.stack([lo & np.uint64(M32), lo >> np.uint64(32), hi & np.uint64(M32), hi >> np.uint64(32)], axis=-1)
  return philox4x32_10(c.astype(np.uint32), key).reshape(-1)[:int(n)]


def _uint32_to_float(x):
  """random_distributions.h Uint32ToFloat: 23 mantissa bits -> [1,2) - 1."""
  v = (np.uint32(127) << np.uint32(23)) | (x & np.uint32(0x7FFFFF))
  return v.view(np.float32) - np.float32(1.0)


def stateless_random_uniform(n, seed0, seed1, minval=0.0, maxval=1.0):
  """tf.random.stateless_uniform float32: rnd * (maxval - minval) + minval."""
  rnd = _uint32_to_float(stateless_u32(n, seed0, seed1))
  return (rnd * np.float32(np.float32(maxval) - np.float32(minval)) + np.float32(minval)).astype(np.float32)


def stateless_random_normal(n, seed0, seed1, mean=0.0, stddev=1.0):
  """tf.random.stateless_normal float32: Box-Muller on consecutive uint32 pairs
  (random_distributions.h BoxMullerFloat), then rnd * stddev + mean."""
  n = int(n)
  u = stateless_u32((n + 3) // 4 * 4, seed0, seed1).reshape(-1, 2)
  u1 = _uint32_to_float(u[:, 0])
  u1 = np.maximum(u1, np.float32(1.0e-7))
  v1 = (np.float32(2.0) * np.float32(np.pi)) * _uint32_to_float(u[:, 1])
  u2 = np.sqrt(np.float32(-2.0) * np.log(u1).astype(np.float32)).astype(np.float32)
  f = np.stack([np.sin(v1).astype(np.float32) * u2, np.cos(v1).astype(np.float32) * u2], axis=-1).reshape(-1)[:n]
  return (f * np.float32(stddev) + np.float32(mean)).astype(np.float32)
