"""The N > 1 path of bench.py on ONE GPU: two ranks share cuda:0 and exchange the
gradient arena through gloo (RCCL refuses two ranks on one device).  Exercises the
real kernels with GradSync's bucketed all-reduce launched from inside backward, the
folded 1/world scale of K3 and a mask update on replica-summed gradients: different
data on the two ranks must leave bit-identical masks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_keep_identical_masks():
  env = dict(os.environ, RIGL_BENCH_ONE_DEVICE='1', RIGL_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', '29591', os.path.join(ROOT, 'bench.py'),
         '--gpus', '2', '--steps', '3', '--warmup', '0', '--batch', '8', '--no-cpu-baseline']
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
  d = json.loads(line)
  assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 16
  assert d['config']['mask_updates_in_timed_region'] == 1          # step 0 is a mask update (begin_step = 0)
  assert d['config']['masks_identical_across_ranks'] is True
  assert d['value'] > 0 and d['roofline']['frac'] > 0
  assert d['allreduce']['bytes'] > 4 * 25_000_000 and d['allreduce']['bus_GBps'] > 0 and d['allreduce']['buckets'] >= 3
