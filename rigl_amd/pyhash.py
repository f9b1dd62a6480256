"""``hash(str)`` as the reference's process computes it.

The reference seeds its stateless noise with Python's salted string hash
(``hash(weights.name + 'drop')``, rigl/sparse_optimizers_base.py:262-270,
:526-534), so a run is reproducible only under a fixed PYTHONHASHSEED
(SURVEY F8).  ``name_hash`` returns the builtin hash when this process was
itself started with an integer PYTHONHASHSEED (then it IS the reference's
value), and otherwise the value a PYTHONHASHSEED=0 process would compute
(CPython 3.4-3.10: SipHash-2-4 with an all-zero key) -- deterministic, and a
legal draw of the reference's behaviour.  ``python_str_hash(text, seed)``
gives the value for any PYTHONHASHSEED without restarting the interpreter.
"""
import os
import struct

_M = (1 << 64) - 1


def _rot(x, b):
  return ((x << b) | (x >> (64 - b))) & _M


def _sipround(v):
  v0, v1, v2, v3 = v
  v0 = (v0 + v1) & _M
  v1 = _rot(v1, 13) ^ v0
  v0 = _rot(v0, 32)
  v2 = (v2 + v3) & _M
  v3 = _rot(v3, 16) ^ v2
  v0 = (v0 + v3) & _M
  v3 = _rot(v3, 21) ^ v0
  v2 = (v2 + v1) & _M
  v1 = _rot(v1, 17) ^ v2
  v2 = _rot(v2, 32)
  return [v0, v1, v2, v3]


def _siphash24(buf, k0, k1):
  v = [k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d,
       k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573]
  tail = len(buf) % 8
  words = struct.unpack('<%dQ' % (len(buf) // 8), buf[:len(buf) - tail])
  last = int.from_bytes(buf[len(buf) - tail:], 'little') | ((len(buf) & 0xFF) << 56)
  for m in words + (last,):
    v[3] ^= m
    v = _sipround(_sipround(v))
    v[0] ^= m
  v[2] ^= 0xFF
  for _ in range(4):
    v = _sipround(v)
  return v[0] ^ v[1] ^ v[2] ^ v[3]


def _secret(seed):
  if seed == 0:
    return 0, 0
  x, raw = seed & 0xFFFFFFFF, bytearray()
  for _ in range(16):                      # k0, k1 = first 16 bytes of _Py_HashSecret (lcg_urandom)
    x = (x * 214013 + 2531011) & 0xFFFFFFFF
    raw.append((x >> 16) & 0xFF)
  return struct.unpack('<2Q', bytes(raw))


def python_str_hash(text, hashseed=0):
  """hash(text) of a CPython (>= 3.4, < 3.11) started with PYTHONHASHSEED=hashseed."""
  if not text:
    return 0
  widest = max(map(ord, text))
  enc = 'latin-1' if widest < 0x100 else ('utf-16-le' if widest < 0x10000 else 'utf-32-le')
  h = _siphash24(text.encode(enc), *_secret(int(hashseed)))
  h = h - (1 << 64) if h >> 63 else h
  return -2 if h == -1 else h


def name_hash(text):
  env = os.environ.get('PYTHONHASHSEED', '')
  if env.isdigit():
    return hash(text)                      # the interpreter already is the reference's configuration
  return python_str_hash(text, 0)
