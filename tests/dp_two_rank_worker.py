#!/usr/bin/env python3
"""Worker of tests/test_dp_two_gpu.py: one rank of a data-parallel run of the MNIST MLP (BASELINE config 1 -- no batch
norm, so a two-replica run on halves of a batch IS the one-replica run on the whole batch up to fp32 reassociation).

  python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_two_rank_worker.py --steps 4 --batch 64 --out f.json
  python tests/dp_two_rank_worker.py --steps 4 --batch 128 --single --out g.json           (the one-rank reference)

Every rank draws the SAME 2 x batch images (seeded), takes its own half, and runs ``SparseRigLOptimizer.minimize`` with
the RCCL gradient exchange (sparse_optimizers_base.py:471-476, imagenet_train_eval.py:363-365).  Step 0 is a mask update
on replica-summed gradients.  Rank 0 writes: the mean over ranks of the per-step losses, a checksum of the masks, whether
all ranks hold identical masks, and the weights' L2 norm."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=4)
  ap.add_argument('--batch', type=int, default=64, help='per-rank batch')
  ap.add_argument('--single', action='store_true', help='one rank on the concatenated batch, no process group')
  ap.add_argument('--one-device-gloo', action='store_true',
                  help='development / one-GPU boxes: all ranks on cuda:0, gradients exchanged through gloo')
  ap.add_argument('--out', required=True)
  a = ap.parse_args()
  world = 1 if a.single else int(os.environ['WORLD_SIZE'])
  rank = 0 if a.single else int(os.environ['RANK'])
  local = 0 if (a.single or a.one_device_gloo) else int(os.environ['LOCAL_RANK'])
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)
  if not a.single:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if a.one_device_gloo:
      dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
      dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.dist import GradSync
  from rigl_amd.workloads import mnist_mlp
  g = V.reset_default_graph(dev)
  model = mnist_mlp.MnistMLP(g, seed=0)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'random', 0.9, {'layer3': 0.0})()
  total = a.batch * (1 if a.single else world)
  gen = torch.Generator(device='cpu').manual_seed(4321)
  images = torch.rand(total, 784, generator=gen).to(torch.bfloat16)
  labels = torch.randint(0, 10, (total,), generator=gen)
  lo = 0 if a.single else rank * a.batch
  x, y = images[lo:lo + a.batch].to(dev), labels[lo:lo + a.batch].to(dev)
  sync = None if a.single else GradSync(g)
  inner = train.MomentumOptimizer(0.2, 0.9, use_nesterov=True, graph=g, grad_sync=sync)
  opt = SO.SparseRigLOptimizer(inner, 0, 50000, 2, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.,
                               use_tpu=sync is not None)
  gs = g.get_or_create_global_step()
  losses = []
  for _ in range(a.steps):
    loss = model.loss(x, y)
    opt.minimize(loss, gs)
    lv = loss.detach().float().reshape(1).clone()
    if sync is not None:
      dist.all_reduce(lv)
      lv /= world
    losses.append(float(lv.item()))
  torch.cuda.synchronize()
  bits = g.BITS.to(torch.int64)
  chk = int((bits * torch.arange(1, bits.numel() + 1, device=dev)).sum().item())
  same = sync.check_masks_identical() if sync is not None else True
  # the exchange really moved bytes between devices: time one all-reduce of the arena
  bus = None
  if sync is not None:
    buf = torch.ones_like(g.G)
    dist.all_reduce(buf)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
      dist.all_reduce(buf)
    e.record()
    torch.cuda.synchronize()
    bus = 2.0 * (world - 1) / world * buf.numel() * 4 * 5 / (s.elapsed_time(e) * 1e-3) / 1e9
  if rank == 0:
    with open(a.out, 'w') as fh:
      json.dump({'world': world, 'backend': None if a.single else dist.get_backend(), 'losses': losses, 'mask_checksum': chk,
                 'masks_identical_across_ranks': bool(same), 'w_norm': float(g.W.double().norm().item()),
                 'mask_ones': [int(m.sum()) for m in g.get_masks()], 'global_step': int(gs.value), 'bus_GBps': bus}, fh)
  if sync is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
