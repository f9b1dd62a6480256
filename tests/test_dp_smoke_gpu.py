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


@pytest.mark.gpu
def test_bench_launches_itself_for_more_than_one_gpu():
  """`python bench.py --gpus 2` with no launcher around it (the form the driver uses): bench.py re-executes itself under
  torch.distributed.run and rank 0 prints the one JSON line (missing #1)."""
  env = dict(os.environ, RIGL_BENCH_ONE_DEVICE='1', RIGL_BENCH_BACKEND='gloo')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
    env.pop(k, None)
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '0', '--batch', '8',
         '--no-cpu-baseline']
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1 and out.stdout.strip().splitlines() == lines
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['masks_identical_across_ranks'] is True


@pytest.mark.gpu
def test_bench_contract_single_gpu():
  """The JSON line the driver parses: contract keys, the roofline and cpu_baseline objects."""
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--prof-every', '2']
  out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1                                   # exactly one JSON line
  assert out.stdout.strip().splitlines() == lines          # ... and nothing else on stdout
  d = json.loads(lines[0])
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert k in d, k
  assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['unit'] == 'images/sec'
  assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
  assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config']
  assert abs(d['value'] - d['config']['global_batch'] * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
  r = d['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in r, k
  assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and r['unit'] == 'TFLOP/s'
  assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0.0 < r['frac'] < r['layerwise_bound']['mfma_only_ms'] / r['layerwise_bound']['ms_per_step']
  # the second timed step of every run is a mask update: K2 is inside the window and on the line (missing #2)
  assert d['config']['mask_updates_in_timed_region'] == 1
  k2 = r['hbm_kernels']['prune_regrow']
  assert k2['updates_timed'] == 1 and k2['ms_per_update'] > 0 and k2['masked_weights'] == 25502912
  assert r['hbm_kernels']['masked_sgd_momentum']['ms_per_step'] > 0
  assert abs(r['algorithmic_gflop_per_image'] - 24.299077632) < 1e-6      # host-side MAC counter == the analytic figure
  c = d['cpu_baseline']
  for k in ('value', 'unit', 'cores', 'kind', 'sample', 'host_cores'):
    assert k in c, k
  assert c['kind'] == 'port' and c['value'] > 0 and 1 <= c['cores'] <= c['host_cores']
  assert c['batch'] == 32 and c['timed_steps'] >= 1 and c['s_per_mask_update'] > 0   # SURVEY 8(d) protocol


@pytest.mark.gpu
@pytest.mark.parametrize('workload,extra', [('mobilenet_v1', 'depthwise_conv'), ('resnet50_erk99', None), ('wrn22', None)])
def test_bench_other_workloads(workload, extra):
  """BASELINE configs 2, 4, 5 through the same bench line (small batch: this checks the plumbing, not the numbers)."""
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '3', '--warmup', '1',
         '--batch', '16', '--prof-every', '1', '--no-cpu-baseline']
  out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
  assert d['value'] > 0 and d['config']['mask_updates_in_timed_region'] == 1
  r = d['roofline']
  assert r['frac'] > 0 and r['hbm_kernels']['prune_regrow']['ms_per_update'] > 0
  if extra:
    assert r['hbm_kernels'][extra]['achieved'] > 0 and r['hbm_kernels'][extra]['launches_per_step'] > 0


@pytest.mark.gpu
def test_rccl_code_path_executes_with_one_rank():
  """The data-parallel machinery on RCCL itself ('nccl' backend), forced on with a single rank (a gpurun box has one GPU and
  RCCL refuses two ranks per device): communicator init, rank-0 state broadcast, bucketed all-reduce launched from inside
  backward, the coalesced last launch (head of the kernel segment + BN / bias segment in one group call), per-bucket
  timeline with device events, the exposed-communication probe and the mask check all run; a one-rank sum leaves the
  gradients unchanged, so the step must equal the plain single-GPU step bit for bit (same loss trajectory)."""
  env = dict(os.environ, RIGL_BENCH_FORCE_SYNC='1', RIGL_BENCH_BACKEND='nccl', MASTER_ADDR='127.0.0.1', MASTER_PORT='29593',
             RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1', '--batch', '16',
         '--no-cpu-baseline']
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  # RCCL prints a version banner through C stdio; bench.py keeps stdout for the JSON line alone (the driver parses it)
  assert len(out.stdout.strip().splitlines()) == 1, out.stdout[-600:]
  d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
  assert d['config']['gradient_exchange'] == 'on (nccl, world 1)'
  assert d['config']['masks_identical_across_ranks'] is True
  ar = d['allreduce']
  assert ar['buckets'] >= 3 and ar['bytes'] > 4 * 25_000_000
  rows = ar['in_step']['buckets']
  assert len(rows) >= 3 and sum(r['bytes'] for r in rows) == ar['bytes']          # every gradient element exchanged exactly once
  assert all('device_ms_after_first_bucket' in r for r in rows) and ar['in_step']['tail_wait_ms'] is not None
  import math
  assert 'comm_exposed_ms' in ar['exposed'] and math.isfinite(float(ar['exposed']['comm_exposed_ms']))
  assert math.isfinite(float(ar['in_step']['tail_wait_ms'])) and float(ar['in_step']['tail_wait_ms']) >= 0.0
  # and the numbers are those of the plain step
  plain = subprocess.run(cmd, env={k: v for k, v in env.items() if k != 'RIGL_BENCH_FORCE_SYNC'}, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
  assert plain.returncode == 0, plain.stderr[-2000:]
  p = json.loads([l for l in plain.stdout.splitlines() if l.startswith('{')][-1])
  assert p['config']['gradient_exchange'] is None and 'allreduce' not in p
