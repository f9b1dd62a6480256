"""The N > 1 path on CPU: world_size-2 `gloo` processes exercise the bucketed
gradient-arena all-reduce (ordering, bucketing, out-of-order notifications,
1/world scaling) and the drop-fraction broadcast.  No device compute."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q, bucket_bytes=4 * 4000):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from rigl_amd import variables as V
    from rigl_amd.dist import GradSync, broadcast_drop_fraction
    g = V.Graph('cpu')
    shapes = [(3, 3, 8, 16), (1, 1, 16, 32), (3, 3, 32, 32), (1, 1, 32, 64), (64, 10)]
    vs = []
    for i, s in enumerate(shapes):
      vs.append(g.add_variable('l%d/weights' % i, s, V.KIND_MASKED if i != 1 else V.KIND_DENSE, 1e-4))
    bn = g.add_variable('bn/gamma', (64,), V.KIND_OTHER)
    g.finalize()
    # tiny buckets so that several are launched mid-"backward"
    sync = GradSync(g, bucket_bytes=bucket_bytes)
    assert sync.world == world and sync.grad_scale == 1.0 / world
    ordered = sorted(vs, key=lambda v: v.offset)
    for trial, order in enumerate([list(reversed(ordered)),                       # normal backward order
                                   [ordered[4], ordered[2], ordered[3], ordered[0], ordered[1]]]):  # out of order
      g.G.zero_()
      for v in order:
        v.grad.fill_(float(rank + 1) * (1 + vs.index(v)))                        # "wgrad"
        sync.notify_layer_grad_ready(v)
      bn.grad.fill_(10.0 * (rank + 1))
      sync.all_reduce(g)
      tot = sum(r + 1 for r in range(world))
      for i, v in enumerate(vs):
        assert torch.all(v.grad == tot * (1 + i)), (trial, i, v.grad.flatten()[:3])
      assert torch.all(bn.grad == 10.0 * tot)
      # alignment padding between tensors stays zero
      used = torch.zeros_like(g.G, dtype=torch.bool)
      for v in vs + [bn]:
        used[v.offset:v.offset + v.numel] = True
      assert torch.all(g.G[~used] == 0)
      if trial == 0:
        assert sync.n_buckets_last >= 3          # really bucketed, not one flush
    f = broadcast_drop_fraction(0.3 if rank == 0 else 0.9)
    assert abs(f - 0.3) < 1e-7
    q.put((rank, 'ok'))
  except Exception as e:  # pylint: disable=broad-except
    import traceback
    q.put((rank, 'FAIL %s\n%s' % (e, traceback.format_exc())))
  finally:
    dist.destroy_process_group()


def test_grad_sync_world2_gloo():
  world = 2
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
  assert all(r[1] == 'ok' for r in res), res


def test_grad_sync_world4_gloo_uneven_bucket_tails():
  """Four ranks, a bucket size that divides neither the arena nor any tensor (1777 elements): the tail buckets are
  ragged and the last launch carries the head of the kernel segment plus the BN / bias segment."""
  world = 4
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q, 4 * 1777)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=180) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
  assert all(r[1] == 'ok' for r in res), res


def test_grad_sync_timeline_records_every_bucket():
  """The per-bucket record bench.py prints: one row per launch, offsets descending (backward order), bytes adding up
  to the whole arena."""
  sys.path.insert(0, ROOT)
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(_free_port())
  dist.init_process_group('gloo', rank=0, world_size=1)
  try:
    from rigl_amd import variables as V
    from rigl_amd.dist import GradSync
    g = V.Graph('cpu')
    vs = [g.add_variable('l%d/weights' % i, (3, 3, 16, 16), V.KIND_MASKED) for i in range(6)]
    bn = g.add_variable('bn/gamma', (64,), V.KIND_OTHER)
    g.finalize()
    s = GradSync(g, bucket_bytes=4 * 3000, enabled=True)      # world 1, exchange forced on: exercises every code path
    s.reset_timeline(device_events=False)
    for v in sorted(vs, key=lambda v: -v.offset):
      v.grad.fill_(1.0)
      s.notify_layer_grad_ready(v)
    bn.grad.fill_(2.0)
    s.all_reduce(g)
    tl = s.timeline_summary()
    rows = tl['buckets']
    assert len(rows) == s.n_buckets_last >= 3
    offs = [r['arena_offset'] for r in rows[:-1]]              # the final row is the BN / bias segment (gloo: its own launch)
    assert offs == sorted(offs, reverse=True) and offs[-1] == 0
    assert rows[-1]['arena_offset'] == g.seg[V.KIND_DENSE][1]
    assert sum(r['bytes'] for r in rows) == 4 * g.G.numel()     # every element exchanged exactly once
    assert torch.all(vs[0].grad == 1.0) and torch.all(bn.grad == 2.0)
  finally:
    dist.destroy_process_group()


def test_grad_sync_single_process_is_noop():
  sys.path.insert(0, ROOT)
  from rigl_amd import variables as V
  from rigl_amd.dist import GradSync
  g = V.Graph('cpu')
  v = g.add_variable('a/weights', (4, 4), V.KIND_MASKED)
  g.finalize()
  s = GradSync(g)
  assert not s.enabled and s.grad_scale == 1.0
  v.grad.fill_(2.0)
  s.notify_layer_grad_ready(v)
  s.all_reduce(g)
  assert torch.all(v.grad == 2.0)


def test_bucket_size_default_and_knob(monkeypatch):
  """GradSync's bucket size: 32 MB unless RIGL_DP_BUCKET_MB says otherwise (an explicit argument wins)."""
  from rigl_amd import variables as V
  from rigl_amd.dist import GradSync
  g = V.Graph('cpu')
  g.add_variable('l/weights', (3, 3, 8, 8), V.KIND_MASKED)
  g.finalize()
  monkeypatch.delenv('RIGL_DP_BUCKET_MB', raising=False)
  assert GradSync(g, enabled=False).bucket_elems == (32 << 20) // 4
  monkeypatch.setenv('RIGL_DP_BUCKET_MB', '8')
  assert GradSync(g, enabled=False).bucket_elems == (8 << 20) // 4
  assert GradSync(g, bucket_bytes=4000, enabled=False).bucket_elems == 1000


def test_bench_self_launch_command():
  """bench.py --gpus N without a launcher re-executes itself under torch.distributed.run (one rank per GPU, 127.0.0.1
  rendezvous, same arguments); with WORLD_SIZE set it is a rank and does not."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, RIGL_BENCH_DRY_LAUNCH='1')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '7', '--warmup', '2'],
                       env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  cmd = json.loads(out.stdout.strip().splitlines()[-1])['launch']
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
  assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
  assert int(cmd[cmd.index('--master-port') + 1]) > 0
  i = cmd.index(os.path.join(root, 'bench.py'))
  assert cmd[i + 1:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
