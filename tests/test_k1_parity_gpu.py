"""K1 parity at the kernel-selection settings that matter, through tests/k1_check.py (a subprocess per setting: the
library reads RIGL_* once per process).  The checker is tests/convref.py's fp64 convolution of the same bf16
operands; bound per element 2^-8 |ref| + 1e-5 sum|a||b| (fwd, dgrad), 1e-5 sum|a||b| (wgrad) -- the north star's
fp32 tolerance relative to the magnitude of the dot product."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=1500):
  e = dict(os.environ)
  e.update(env or {})
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'k1_check.py')] + args, env=e,
                     capture_output=True, text=True, timeout=timeout)
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert lines, 'no verdict line; stderr tail: %s' % r.stderr[-3000:]
  out = json.loads(lines[-1])
  assert r.returncode == 0 and out['ok'], '%s\n%s' % (out, r.stdout[-3000:])
  return out


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [
    {'RIGL_T196': '2', 'RIGL_C3': '2', 'RIGL_W9': '1'},     # every new kernel wherever its shape is legal (fwd, dgrad, wgrad)
    {'RIGL_T196': '1', 'RIGL_C3': '1'},                     # the forward-only selection rules
    {},                                                     # the defaults
])
def test_tile196_shapes(env):
  out = _run(['--set', 't196'], env)
  assert out['cases'] == 12


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('env', [{}, {'RIGL_T196': '2', 'RIGL_C3': '2'}])
def test_resnet50_layer_shapes_at_batch_128(env):
  """All distinct ResNet-50 conv shapes at the benchmarked per-GPU batch: fwd, fwd + statistics,
  dgrad, dgrad + addend, wgrad and the one-call backward, under the default kernel selection and with the tile196 / 3x3
  slab kernels forced wherever legal -- so the 256x128 forward tile, the parity-class strided dgrad, tile196, the slab
  kernel and the shared-launch split plans are each pinned directly against the fp64 reference at the benchmarked sizes."""
  out = _run(['--set', 'resnet50', '--batch', '128'], env, timeout=3000)
  assert out['cases'] == 23
