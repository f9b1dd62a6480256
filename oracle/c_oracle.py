"""ctypes wrapper of oracle/c/librigl_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c')
_lib = None


def load():
  global _lib
  if _lib is None:
    so = os.path.join(_DIR, 'librigl_oracle.so')
    if not os.path.exists(so):
      subprocess.check_call(['make', '-C', _DIR])
    _lib = C.CDLL(so)
    _lib.rigl_oracle_update.restype = C.c_int
  return _lib


def update(score_drop, score_grow, mask, w, drop_fraction, momentum=None, grow_values=None, momentum_values=None,
           reinit_when_same=False):
  """Returns (new_mask, new_w, new_momentum, counts)."""
  f = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1)).copy()
  sd, sg, m, w2, mom, gv, mv = map(f, (score_drop, score_grow, mask, w, momentum, grow_values, momentum_values))
  n = sd.size
  new_mask = np.zeros(n, np.float32)
  counts = np.zeros(5, np.int32)
  p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
  rc = load().rigl_oracle_update(C.c_int64(n), p(sd), p(sg), p(m), p(w2), p(mom), p(gv), p(mv),
                                 C.c_float(drop_fraction), C.c_int(int(reinit_when_same)), p(new_mask), p(counts))
  assert rc == 0
  return new_mask, w2, mom, counts
