#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ResNet-50 80 %-ERK RigL training step
(BASELINE.json metric) on N MI355X of one node, synthetic ImageNet-shaped data.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one call of ``SparseRigLOptimizer.minimize`` on a per-GPU batch of
128 images: masked-conv forward + backward (HIP MFMA kernels), DP all-reduce of
the dense gradient arena (RCCL), then either the fused masked Nesterov update
(K3) or -- every 100th step -- the fused prune/regrow mask update (K2).
Nothing is skipped inside the timed region; inputs are resident in HBM.

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     achieved dense-equivalent conv TFLOP/s vs the 2.5 PF bf16 MFMA
                  peak, from HIP events recorded around every K1 launch on the
                  launch stream during the timed region (rigl_prof_*; every 10th step by
                  default -- the stamped dispatches themselves cost queue time);
  "cpu_baseline": the fp32 CPU restatement of the same step (oracle/, "port"),
                  timed on a bounded sample on this host's cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--batch', type=int, default=128, help='per-GPU batch')
  ap.add_argument('--sparsity', type=float, default=0.8)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-prof', action='store_true')
  ap.add_argument('--prof-every', type=int, default=10,
                  help='HIP-event timing of the K1 launches on every Nth timed step (1 = every step); the stamped '
                       'dispatches cost ~4 us each in the queue, 0.7 ms per fully profiled step')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # RIGL_BENCH_ONE_DEVICE=1 + RIGL_BENCH_BACKEND=gloo: development smoke of the N > 1 code path on a
  # single-GPU box (all ranks on cuda:0, gradients exchanged through gloo); never used for numbers.
  one_device = os.environ.get('RIGL_BENCH_ONE_DEVICE', '0') == '1'
  backend = os.environ.get('RIGL_BENCH_BACKEND', 'nccl')
  dev_index = 0 if one_device else local_rank
  torch.cuda.set_device(dev_index)
  dev = torch.device('cuda', dev_index)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(backend)
  assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

  import numpy as np
  from rigl_amd import ops, sparse_optimizers, sparse_utils, train, variables
  from rigl_amd.dist import GradSync
  from rigl_amd.workloads import resnet50, shapes

  # ---- build: ResNet-50, all 54 kernels masked, ERK(-kernel) 0.8 (README.md:84-90)
  g = variables.reset_default_graph(dev)
  model = resnet50.ResNet50(g, seed=0)
  np.random.seed(0)                       # identical masks on every rank
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', args.sparsity, {})()
  sync = GradSync(g) if world > 1 else None
  global_batch = args.batch * world
  lr = 0.1 * global_batch / 256.0          # imagenet_train_eval.py:317-330 (peak LR)
  inner = train.MomentumOptimizer(lr, 0.9, use_nesterov=True, graph=g, grad_sync=sync)
  opt = sparse_optimizers.SparseRigLOptimizer(
      inner, begin_step=0, end_step=25000, frequency=100, drop_fraction=0.3,
      drop_fraction_anneal='cosine', grow_init='zeros', initial_acc_scale=0.0,
      use_tpu=world > 1)
  gs = g.get_or_create_global_step()
  images, labels = resnet50.synthetic_batch(args.batch, dev, seed=1234 + rank)

  def step():
    loss = model.loss(images, labels, label_smoothing=0.1)
    opt.minimize(loss, gs)
    return loss

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    step()
  fence()
  prof_every = max(int(args.prof_every), 1)
  if not args.no_prof:
    ops.prof_collect()
  t0 = time.perf_counter()
  n_updates = 0
  n_profiled = 0
  for i in range(args.steps):
    if not args.no_prof:
      on = i % prof_every == 0 or int(gs.value) % 100 == 0     # ... and the mask-update step (K2)
      ops.prof_enable(on)                  # a flag flip; the events are read after the timed region
      n_profiled += int(on)
    before = gs.value
    step()
    n_updates += int(gs.value == before)
  fence()
  dt = time.perf_counter() - t0
  prof = None
  if not args.no_prof:
    ops.prof_enable(False)
    prof = ops.prof_collect()
  masks_same = sync.check_masks_identical() if sync is not None else None   # after the timed region: replicas must agree
  # C1 on its own, after the timed region: the gradient arena all-reduced in GradSync's buckets, nothing
  # to overlap with -- what the step would pay if none of it hid behind the backward pass.
  ar = None
  if sync is not None:
    buf = torch.zeros_like(g.G)
    be = sync.bucket_elems
    chunks = [buf[i:i + be] for i in range(0, buf.numel(), be)]
    for rep in range(6):
      if rep == 1:                          # first repetition warms the communicator up
        fence()
        ta = time.perf_counter()
      for h in [dist.all_reduce(c, async_op=True) for c in chunks]:
        h.wait()
    fence()
    t_ar = (time.perf_counter() - ta) / 5
    nbytes = buf.numel() * 4
    ar = {'bytes': nbytes, 'buckets': len(chunks), 'ms': t_ar * 1e3,
          'bus_GBps': 2.0 * (world - 1) / world * nbytes / t_ar / 1e9, 'peak_GBps': 7 * 153.0,
          'note': 'fp32 gradient arena, bucketed all-reduce alone (outside the timed region); bus = 2(N-1)/N x bytes / time, '
                  'peak = 7 xGMI links x 153 GB/s per GPU'}
    del buf, chunks
  t = torch.tensor([dt], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  dt = float(t.item())
  ms_per_step = dt / args.steps * 1e3
  value = global_batch * args.steps / dt

  if rank == 0:
    fwd_macs, dgrad_macs = shapes.resnet50_macs_per_image()
    flops_per_step = 2.0 * (2 * fwd_macs + dgrad_macs) * args.batch   # per GPU, dense-equivalent
    out = {
        'metric': 'images/sec/node, ResNet-50 80% ERK RigL step',
        'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'ResNet-50 v1.5, ERK(kernel) %.2f sparse, 54 masked tensors, RigL dT=100 '
                               'drop 0.3 cosine, Nesterov 0.9, wd 1e-4, label smoothing 0.1, 224x224x3 NHWC'
                               % args.sparsity,
                   'global_batch': global_batch, 'per_gpu_batch': args.batch,
                   'parallelism': 'dp%d' % world, 'mask_updates_in_timed_region': n_updates,
                   'masks_identical_across_ranks': masks_same},
    }
    if ar is not None:
      out['allreduce'] = ar
    if prof is not None:
      conv_ms = sum(prof[k][0] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'conv_bwd'))
      launches = sum(prof[k][1] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'conv_bwd'))
      n_fwd_bwd = max(n_profiled, 1)       # every profiled step (update or not) runs fwd + bwd
      achieved = flops_per_step * n_fwd_bwd / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
      # HBM bytes per K1 launch from the PMC passes (rocprofv3 cannot run inside this process: the two
      # counters need separate passes) -- tools/pmc_summary.py writes the figure next to the profiles.
      traffic = None
      try:
        with open(os.path.join(ROOT, 'profiles', 'r1', 'k1_traffic.json')) as fh:
          traffic = float(json.load(fh)['bytes_per_launch'])
      except (OSError, ValueError, KeyError):
        pass
      out['roofline'] = {
          'bound': 'mfma', 'achieved': achieved, 'peak': 2500.0, 'unit': 'TFLOP/s',
          'frac': achieved / 2500.0, 'traffic': traffic,
          'traffic_unit': 'HBM bytes per K1 launch (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes, profiles/r1/pmc_hbm_traffic.csv)',
          'kernel': 'K1 masked conv implicit-GEMM (fwd, dgrad, wgrad; conv_bwd = dgrad + wgrad sharing one launch): all %d '
                    'kernel dispatches of %d of the %d timed steps (every %d-th), each stamped by its own dispatch '
                    '(hipExtLaunchKernelGGL start/stop events)' % (launches, n_profiled, args.steps, prof_every),
          'profiled_steps': n_profiled,
          'algorithmic_gflop_per_image': flops_per_step / args.batch / 1e9,
          'avg_launch_ms': conv_ms / max(launches, 1),
          'conv_ms_per_step': conv_ms / n_fwd_bwd,
          'by_kind_ms_per_step': {k: prof[k][0] / n_fwd_bwd for k in prof},
          'conv_share_of_step': (conv_ms / n_fwd_bwd) / ms_per_step,
      }
      # what the same three GEMMs per layer could reach: each layer at the better of its MFMA and HBM bounds
      lb, lb_mfma, lb_hbm = shapes.resnet50_layerwise_bound(args.batch)
      out['roofline']['layerwise_bound'] = {
          'ms_per_step': lb * 1e3, 'mfma_only_ms': lb_mfma * 1e3, 'hbm_only_ms': lb_hbm * 1e3,
          'frac_of_bound': lb * 1e3 / (conv_ms / n_fwd_bwd) if conv_ms > 0 else 0.0,
          'note': 'sum over layers and fwd/dgrad/wgrad of max(flops / 2.5 PF, algorithmic bytes / 8 TB/s); '
                  'mfma_only_ms / ms_per_step is the highest `frac` any implementation of these layers can reach'}
      # the two HBM-bound kernels of the path, against the 8 TB/s HBM3E peak (algorithmic bytes: SURVEY 8d)
      n_params = sum(v.numel for v in g.trainable_variables())
      n_masked = sum(m.numel for m in g.get_masks())
      k3_ms, k3_n = prof['sgd_momentum']
      k2_ms, k2_n = prof['prune_regrow']
      hbm = {}
      if k3_ms > 0:
        steps_k3 = max(n_profiled - (1 if k2_n else 0), 1)      # update iterations skip the weight update (F9)
        gbs = 20.1 * n_params / (k3_ms / steps_k3 * 1e-3) / 1e9
        hbm['masked_sgd_momentum'] = {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                                      'algorithmic_bytes_per_param': 20.1, 'ms_per_step': k3_ms / steps_k3}
      if k2_ms > 0 and k2_n > 0:
        gbs = 8.25 * n_masked / (k2_ms / k2_n * 1e-3) / 1e9
        hbm['prune_regrow'] = {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                               'algorithmic_bytes_per_weight': 8.25, 'ms_per_update': k2_ms / k2_n}
      out['roofline']['hbm_kernels'] = hbm
    if world == 1 and not args.no_cpu_baseline:
      try:
        from oracle import resnet_cpu
        out['cpu_baseline'] = resnet_cpu.time_cpu_baseline(batch=8, steps=2)
      except Exception as e:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': 'failed: %r' % (e,)}
    print(json.dumps(out))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
