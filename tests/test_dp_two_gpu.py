"""Two ranks on TWO devices over RCCL ('nccl'): skipped wherever fewer than two GPUs are visible (the gpurun boxes have
one), so that the first multi-GPU node that runs `pytest -m gpu` produces the evidence without anyone in the loop
(VERDICT r3, next #6).  Reference: the cross-replica gradient sum of sparse_optimizers_base.py:471-476 and
imagenet_train_eval.py:363-365 (CrossShardOptimizer)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip('torch')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
two_gpus = pytest.mark.skipif(N_DEV < 2, reason='needs two GPUs (found %d)' % N_DEV)


def _env():
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'RIGL_BENCH_ONE_DEVICE', 'RIGL_BENCH_BACKEND',
            'RIGL_BENCH_FORCE_SYNC'):
    env.pop(k, None)
  return env


@pytest.mark.gpu
@two_gpus
def test_bench_two_ranks_over_rccl():
  """`python bench.py --gpus 2` bare: it re-executes itself under torch.distributed.run, one rank per GPU; the gradient
  arena is all-reduced over RCCL from inside backward and both ranks derive the same masks from different data."""
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--batch', '32',
         '--no-cpu-baseline']
  out = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1 and out.stdout.strip().splitlines() == lines
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 64
  assert d['config']['gradient_exchange'] == 'on (nccl, world 2)'
  assert d['config']['masks_identical_across_ranks'] is True
  assert d['config']['mask_updates_in_timed_region'] == 1
  ar = d['allreduce']
  assert ar['bus_GBps'] > 0 and ar['buckets'] >= 3 and ar['bytes'] > 4 * 25_000_000
  rows = ar['in_step']['buckets']
  assert sum(r['bytes'] for r in rows) == ar['bytes']            # every gradient element exchanged exactly once
  # the overlap evidence an 8-GPU node produces with nobody in the loop: exposed exchange time per step and the
  # compute-stream wait behind backward for the last bucket -- present and finite
  import math
  assert 'comm_exposed_ms' in ar['exposed'] and math.isfinite(float(ar['exposed']['comm_exposed_ms']))
  assert 'tail_wait_ms' in ar['in_step'] and ar['in_step']['tail_wait_ms'] is not None
  assert math.isfinite(float(ar['in_step']['tail_wait_ms'])) and float(ar['in_step']['tail_wait_ms']) >= 0.0
  assert d['value'] > 0 and d['scaling'] == 'weak' and d['roofline']['frac'] > 0


def _two_vs_one(tmp_path, extra, backend, port):
  worker = os.path.join(ROOT, 'tests', 'dp_two_rank_worker.py')
  f2, f1 = str(tmp_path / 'two.json'), str(tmp_path / 'one.json')
  cmd2 = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
          '--master-port', str(port), worker, '--steps', '5', '--batch', '64', '--out', f2] + extra
  out = subprocess.run(cmd2, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  cmd1 = [sys.executable, worker, '--steps', '5', '--batch', '128', '--single', '--out', f1]
  out = subprocess.run(cmd1, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  with open(f2) as fh:
    two = json.load(fh)
  with open(f1) as fh:
    one = json.load(fh)
  assert two['world'] == 2 and two['backend'] == backend and two['masks_identical_across_ranks'] is True
  assert two['bus_GBps'] > 0
  assert two['global_step'] == one['global_step'] and two['mask_ones'] == one['mask_ones']
  # step 0 has identical weights: the mean loss differs by fp32 reassociation only.  Later steps: the replica-summed
  # gradient differs from the one-rank gradient in its last fp32 bits, and a weight that lands on the other side of a
  # bf16 rounding boundary of the operand shadow moves the loss by ~1e-5 (measured 1.5e-5 after three updates) -- 1e-4.
  assert abs(two['losses'][0] - one['losses'][0]) <= 1e-6 * abs(one['losses'][0]), (two['losses'], one['losses'])
  for a, b in zip(two['losses'], one['losses']):
    assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (two['losses'], one['losses'])
  assert abs(two['w_norm'] - one['w_norm']) <= 1e-5 * one['w_norm']


@pytest.mark.gpu
@two_gpus
def test_two_replicas_equal_one_replica_on_the_concatenated_batch(tmp_path):
  """MNIST MLP (no batch norm): two RCCL ranks on halves of a batch vs one rank on the whole batch -- the mean loss of
  every step within fp32 reassociation of the gradients (first step 1e-6, later steps 1e-4 relative: bf16 operand re-rounding), the same number of connections per layer, identical masks on
  both ranks.  Steps 0 and 2 are mask updates on the replica-summed dense gradients."""
  _two_vs_one(tmp_path, [], 'nccl', 29597)


@pytest.mark.gpu
def test_two_replicas_equal_one_replica_one_device_gloo(tmp_path):
  """The same statement on a one-GPU box: both ranks on cuda:0, gradients through gloo (RCCL refuses two ranks per device).
  Pins the test's own arithmetic -- what differs on a two-GPU node is only the transport."""
  _two_vs_one(tmp_path, ['--one-device-gloo'], 'gloo', 29598)
