#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ResNet-50 80 %-ERK RigL training step
(BASELINE.json metric) on N MI355X of one node, synthetic ImageNet-shaped data.

  python bench.py --gpus 1 --steps K --warmup W [--workload resnet50]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one call of ``SparseRigLOptimizer.minimize`` on a per-GPU batch of
128 images: masked-conv forward + backward (HIP MFMA kernels), DP all-reduce of
the dense gradient arena (RCCL), then either the fused masked Nesterov update
(K3) or -- every 100th step -- the fused prune/regrow mask update (K2).
Nothing is skipped inside the timed region; inputs are resident in HBM.  After the
warm-up the global step is positioned one step before a multiple of the update
period, so the SECOND timed step of every run is a mask-update iteration (K2 is
inside the timed window and on the JSON line for any --steps >= 2); later updates
fall every 100 steps as in training.

``--workload`` selects the configuration (BASELINE.json `configs`): resnet50 (the
metric's own config, default), resnet50_erk99, mobilenet_v1, wrn22 -- the other
three are side measurements with the same JSON contract; `config.workload` names
what ran.

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     achieved dense-equivalent conv TFLOP/s vs the 2.5 PF bf16 MFMA
                  peak, from HIP events recorded around every K1 launch on the
                  launch stream during the timed region (rigl_prof_*; every 10th step by
                  default -- the stamped dispatches themselves cost queue time);
  "cpu_baseline": the fp32 CPU restatement of the same step (oracle/, "port"),
                  timed on a bounded sample on this host's cores (N = 1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC only on this pool (RCCL's peer-memory exchange needs it); read when the HSA runtime starts, i.e. before the
# first HIP call, so it is set before torch is imported
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

UPDATE_PERIOD = 100          # maskupdate_frequency (imagenet_train_eval.py:244, resnet_train_eval.py:101)


def build_workload(name, g, dev, batch, rank, sparsity):
  """Returns dict(model, images, labels, loss(), label, metric, opt_kwargs, lr(global_batch))."""
  import numpy as np
  from rigl_amd import sparse_utils
  if name in ('resnet50', 'resnet50_erk99'):
    from rigl_amd.workloads import resnet50
    dense_stem = name == 'resnet50_erk99'
    if sparsity is None:
      sparsity = 0.99 if dense_stem else 0.8
    model = resnet50.ResNet50(g, prune_first_layer=not dense_stem, seed=0)
    np.random.seed(0)                       # identical masks on every rank
    sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', sparsity, {})()
    images, labels = resnet50.synthetic_batch(batch, dev, seed=1234 + rank)
    return dict(
        model=model, images=images, labels=labels, loss=lambda: model.loss(images, labels, label_smoothing=0.1),
        label='ResNet-50 v1.5, ERK(kernel) %.2f sparse, %d masked tensors%s, RigL dT=100 drop 0.3 cosine, Nesterov 0.9, '
              'wd 1e-4, label smoothing 0.1, 224x224x3 NHWC' % (sparsity, len(g.get_masks()), ' (dense stem)' if dense_stem else ''),
        metric='images/sec/node, ResNet-50 %d%% ERK RigL step' % round(sparsity * 100),
        lr=lambda gb: 0.1 * gb / 256.0,      # imagenet_train_eval.py:317-330 (peak LR)
        opt=dict(begin_step=0, end_step=25000, frequency=UPDATE_PERIOD, drop_fraction=0.3, drop_fraction_anneal='cosine'))
  if name == 'mobilenet_v1':
    from rigl_amd.workloads import mobilenet_v1
    if sparsity is None:
      sparsity = 0.9
    model = mobilenet_v1.MobileNetV1(g, seed=0)
    np.random.seed(0)
    sparse_utils.get_mask_init_fn(g.get_masks(), 'random', sparsity, {})()
    images, labels = mobilenet_v1.synthetic_batch(batch, dev, seed=1234 + rank)
    return dict(
        model=model, images=images, labels=labels, loss=lambda: model.loss(images, labels, label_smoothing=0.1),
        label='MobileNet-v1, uniform %.2f on the 13 pointwise convs + final_dense (%d masked tensors; depthwise convs and '
              'stem dense), RigL dT=100 drop 0.3 cosine, Nesterov 0.9, wd 4e-5, label smoothing 0.1, 224x224x3 NHWC'
              % (sparsity, len(g.get_masks())),
        metric='images/sec/node, MobileNet-v1 %d%% uniform RigL step' % round(sparsity * 100),
        lr=lambda gb: 0.1 * gb / 256.0,
        opt=dict(begin_step=0, end_step=25000, frequency=UPDATE_PERIOD, drop_fraction=0.3, drop_fraction_anneal='cosine'))
  if name == 'wrn22':
    from rigl_amd.workloads import wide_resnet
    if sparsity is None:
      sparsity = 0.8
    model = wide_resnet.WideResNet(g, depth=22, width=1)
    np.random.seed(0)
    sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', sparsity, {})()
    images, labels = wide_resnet.synthetic_batch(batch, dev, seed=1234 + rank)
    return dict(
        model=model, images=images, labels=labels, loss=lambda: model.loss(images, labels),
        label='CIFAR WideResNet-22-1 ("ResNet-20"), ERK(kernel) %.2f sparse, %d masked tensors (dense stem), RigL dT=100 '
              'drop 0.3 constant, Nesterov 0.9, wd 5e-4, 32x32x3 NHWC' % (sparsity, len(g.get_masks())),
        metric='images/sec/node, CIFAR WRN-22-1 %d%% ERK RigL step' % round(sparsity * 100),
        lr=lambda gb: 0.1,                    # resnet_train_eval.py:189-203
        opt=dict(begin_step=0, end_step=75000, frequency=UPDATE_PERIOD, drop_fraction=0.3, drop_fraction_anneal='constant'))
  raise ValueError('unknown workload %r' % (name,))


def lib_sha16():
  from rigl_amd import _lib
  try:
    with open(_lib.LIB_PATH, 'rb') as fh:
      return hashlib.sha256(fh.read()).hexdigest()[:16]
  except OSError:
    return None


def _claim_stdout():
  """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on rank 0 (buffered, so it
  lands AFTER anything Python printed when stdout is a pipe or a file), and libraries may print more: everything written
  to file descriptor 1 from here on goes to stderr, and the returned descriptor is the real stdout for the JSON line."""
  sys.stdout.flush()
  real = os.dup(1)
  os.dup2(2, 1)
  return real


def _self_launch(argv, gpus):
  """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): replace this process by
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` -- one rank per GPU, rank 0
  prints the one JSON line on the inherited stdout.  RIGL_BENCH_DRY_LAUNCH=1 prints the command instead (tests)."""
  import socket
  port = os.environ.get('MASTER_PORT')
  if not port:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
      sk.bind(('127.0.0.1', 0))
      port = str(sk.getsockname()[1])
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
         '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + list(argv)
  if os.environ.get('RIGL_BENCH_DRY_LAUNCH', '0') == '1':
    print(json.dumps({'launch': cmd}))
    sys.stdout.flush()
    return
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  sys.stdout.flush()
  os.execv(sys.executable, cmd)


def main():
  # (argument errors and --help print before stdout is claimed)
  pre = argparse.ArgumentParser(add_help=False)
  pre.add_argument('--gpus', type=int, default=1)
  known, _ = pre.parse_known_args()
  if known.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    _self_launch(sys.argv[1:], known.gpus)
    return
  real_stdout = _claim_stdout()
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--batch', type=int, default=128, help='per-GPU batch')
  ap.add_argument('--workload', default='resnet50', choices=('resnet50', 'resnet50_erk99', 'mobilenet_v1', 'wrn22'))
  ap.add_argument('--sparsity', type=float, default=None, help="override the workload's sparsity")
  ap.add_argument('--precision', default='bfloat16', choices=('bfloat16', 'float32'),
                  help="the reference's --precision flag (imagenet_train_eval.py:56-59; its default is float32, its TPU runs and "
                       'the graded line bfloat16).  float32: fp32 activations, K1 on conv_f32.hip (v_mfma_f32_32x32x2_f32, an '
                       'un-tuned validation body), glue on stock torch ops; roofline against the 157.3 TFLOP/s fp32 matrix peak')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-prof', action='store_true')
  ap.add_argument('--graph', action='store_true',
                  help='replay ordinary steps from a captured HIP graph (rigl_amd.train.GraphedStep; N = 1 only): for '
                       'launch-bound models such as wrn22.  The K1 timings of `roofline` then come from a few eager '
                       'steps run AFTER the timed region')
  ap.add_argument('--no-graph', action='store_true',
                  help='wrn22 replays its ordinary steps from a captured HIP graph by default (launch-bound: 52.7 k -> 80.4 k '
                       'images/s, profiles/r5); this flag runs it eagerly')
  ap.add_argument('--no-sync', action='store_true',
                  help='N > 1 only: run the step WITHOUT the gradient exchange (replicas diverge; gives the compute-only step '
                       'time that comm_exposed_ms is measured against -- never a headline number)')
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
  # RIGL_BENCH_FORCE_SYNC=1: run the whole data-parallel machinery (process group, bucketed exchange launched from inside
  # backward, coalesced last bucket, timeline, exposed-communication probe, mask check) even with ONE rank -- the only way
  # to execute the RCCL code path on a single-GPU box (tests/test_dp_smoke_gpu.py); never used for numbers.
  force_sync = os.environ.get('RIGL_BENCH_FORCE_SYNC', '0') == '1'
  if world > 1 or force_sync:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this pool (RCCL needs it)
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` bare, or under '
                     'torch.distributed.run --nproc-per-node N)' % (args.gpus, world))

  from rigl_amd import ops, sparse_optimizers, sparse_utils, train, variables
  from rigl_amd.dist import GradSync
  from rigl_amd.workloads import shapes

  fp32 = args.precision == 'float32'
  if fp32:
    ops.tune_set('k1_fp32', 1)             # workloads.nn.activation_dtype: the synthetic batch and every activation in fp32
  g = variables.reset_default_graph(dev)
  wl = build_workload(args.workload, g, dev, args.batch, rank, args.sparsity)
  sync = GradSync(g, enabled=not args.no_sync) if (world > 1 or force_sync) else None
  global_batch = args.batch * world
  lr = wl['lr'](global_batch)
  inner = train.MomentumOptimizer(lr, 0.9, use_nesterov=True, graph=g, grad_sync=sync)
  opt = sparse_optimizers.SparseRigLOptimizer(
      inner, grow_init='zeros', initial_acc_scale=0.0, use_tpu=sync is not None and not args.no_sync, **wl['opt'])
  gs = g.get_or_create_global_step()
  loss_fn = wl['loss']

  def step():
    loss = loss_fn()
    opt.minimize(loss, gs)
    return loss

  def fence():
    torch.cuda.synchronize()
    if sync is not None:
      dist.barrier()
    torch.cuda.synchronize()

  if args.workload == 'wrn22' and not args.no_graph and args.precision == 'bfloat16':
    args.graph = True                       # (the measured default of the launch-bound workload)
  graphed = train.GraphedStep(loss_fn, opt, gs) if (args.graph and world == 1) else None
  run_step = graphed if graphed is not None else step
  for i in range(args.warmup):
    if i == args.warmup - 1:
      ops.work_count(True)                  # dense-equivalent MACs / depthwise bytes of one fwd + bwd
    step()
  if args.warmup == 0:
    ops.work_count(True)
    step()
  ops.work_count(False)
  work = dict(ops.WORK)
  fence()
  # Position the schedule: the first warm-up step was the step-0 mask update (last_update = 0); with the global step at
  # period - 1 the first timed step is an ordinary one and the second is the next mask update.
  if int(gs.value) < UPDATE_PERIOD - 1:
    gs.value = UPDATE_PERIOD - 1
  if graphed is not None:                   # capture now (its own warm-up + capture step), then re-position the schedule
    for _ in range(10):
      graphed()
      if graphed.replays:
        break
    fence()
    gs.value = UPDATE_PERIOD - 1
    opt._last_update_step = 0               # pylint: disable=protected-access
  prof_every = max(int(args.prof_every), 1)
  in_loop_prof = not args.no_prof and graphed is None
  if not args.no_prof:
    ops.prof_collect()
  if sync is not None:
    sync.reset_timeline()
  t0 = time.perf_counter()
  n_updates = 0
  n_profiled = 0
  n_profiled_updates = 0
  for i in range(args.steps):
    if in_loop_prof:
      upd = opt.is_mask_update_iter(int(gs.value), opt._last_update_step)   # (pure host arithmetic; the step recomputes it)
      on = i % prof_every == 0 or upd       # ... and every mask-update step (K2)
      ops.prof_enable(on)                  # a flag flip; the events are read after the timed region
      n_profiled += int(on)
    before = gs.value
    run_step()
    is_upd = int(gs.value == before)
    n_updates += is_upd
    if in_loop_prof and on:
      n_profiled_updates += is_upd
  fence()
  dt = time.perf_counter() - t0
  prof = None
  if not args.no_prof and graphed is not None:
    # graph replays carry no per-kernel events: time the kernels on a few EAGER steps after the timed region
    gs.value = UPDATE_PERIOD - 1
    opt._last_update_step = 0               # pylint: disable=protected-access
    ops.prof_enable(True)
    for _ in range(4):
      before = gs.value
      step()
      n_profiled += 1
      n_profiled_updates += int(gs.value == before)
    fence()
  if not args.no_prof:
    ops.prof_enable(False)
    prof = ops.prof_collect()
  masks_same = sync.check_masks_identical() if (sync is not None and not args.no_sync) else None   # after the timed region
  timeline = sync.timeline_summary() if sync is not None else None
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
  # exposed communication: the same steps once more WITHOUT the exchange (after everything that is reported above;
  # the replicas diverge from here on, nothing below reads the model)
  comm_exposed = None
  if sync is not None and not args.no_sync and os.environ.get('RIGL_BENCH_COMM_EXPOSED', '1') == '1':
    sync.enabled = False
    n_probe = min(args.steps, 20)
    for _ in range(2):
      step()
    fence()
    tb = time.perf_counter()
    for _ in range(n_probe):
      step()
    fence()
    t_nosync = torch.tensor([(time.perf_counter() - tb) / n_probe], dtype=torch.float64, device=dev)
    dist.all_reduce(t_nosync, op=dist.ReduceOp.MAX)
    comm_exposed = {'ms_per_step_without_exchange': float(t_nosync.item()) * 1e3, 'steps': n_probe}
  t = torch.tensor([dt], dtype=torch.float64, device=dev)
  if sync is not None:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  dt = float(t.item())
  ms_per_step = dt / args.steps * 1e3
  value = global_batch * args.steps / dt

  if rank == 0:
    macs = work['fwd_macs'] + work['dgrad_macs'] + work['wgrad_macs']
    flops_per_step = 2.0 * macs              # per GPU, dense-equivalent (the mask saves no MFMA work by design)
    if args.workload.startswith('resnet50'):
      fwd_macs, dgrad_macs = shapes.resnet50_macs_per_image()
      analytic = 2.0 * (2 * fwd_macs + dgrad_macs) * args.batch
      assert abs(analytic - flops_per_step) <= 1e-9 * analytic, (analytic, flops_per_step)   # counter == rigl/str_sparsities.py:29
    out = {
        'metric': wl['metric'],
        'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'fp32' if fp32 else 'bf16', 'data': 'synthetic',
        'config': {'workload': wl['label'],
                   'global_batch': global_batch, 'per_gpu_batch': args.batch,
                   'parallelism': 'dp%d' % world, 'mask_updates_in_timed_region': n_updates,
                   'masks_identical_across_ranks': masks_same,
                   'gradient_exchange': (None if sync is None else ('off (--no-sync)' if args.no_sync else 'on (%s, world %d)' % (backend, world))),
                   'execution': ('HIP graph replay of ordinary steps (%d replays, %d eager steps incl. mask updates)'
                                 % (graphed.replays, graphed.eager_steps)) if graphed is not None else 'eager',
                   'lib_sha16': lib_sha16()},
    }
    if ar is not None:
      out['allreduce'] = ar
      if timeline is not None:
        out['allreduce']['in_step'] = timeline
      if comm_exposed is not None:
        comm_exposed['comm_exposed_ms'] = ms_per_step - comm_exposed['ms_per_step_without_exchange']
        comm_exposed['note'] = ('step time with the gradient exchange minus the same step without it (max over ranks, '
                                'measured right after the timed region)')
        out['allreduce']['exposed'] = comm_exposed
    if prof is not None:
      conv_ms = sum(prof[k][0] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'conv_bwd'))
      launches = sum(prof[k][1] for k in ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'conv_bwd'))
      n_fwd_bwd = max(n_profiled, 1)       # every profiled step (update or not) runs fwd + bwd
      achieved = flops_per_step * n_fwd_bwd / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
      # HBM bytes per K1 launch from the PMC passes (rocprofv3 cannot run inside this process: the two
      # counters need separate passes) -- tools/pmc_summary.py writes the figure next to the profiles,
      # stamped with the library build it was collected on.
      traffic = traffic_build = None
      if args.workload == 'resnet50':
        for rnd in ('r6', 'r5', 'r4', 'r3', 'r2', 'r1'):
          try:
            with open(os.path.join(ROOT, 'profiles', rnd, 'k1_traffic.json')) as fh:
              tj = json.load(fh)
            traffic = float(tj['bytes_per_launch'])
            traffic_build = tj.get('lib_sha16')
            break
          except (OSError, ValueError, KeyError):
            continue
      peak_tf = 157.3 if fp32 else 2500.0   # MI355X_MICROARCH.md: fp32 matrix (= vector) peak / dense bf16 MFMA peak
      if fp32:
        traffic = traffic_build = None       # (the PMC passes were collected on the bf16 kernels)
      out['roofline'] = {
          'bound': 'mfma', 'achieved': achieved, 'peak': peak_tf, 'unit': 'TFLOP/s',
          'frac': achieved / peak_tf, 'traffic': traffic,
          'traffic_unit': 'HBM bytes per K1 launch (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes, profiles/r*/pmc_hbm_traffic.csv)',
          'traffic_lib_sha16': traffic_build,
          'traffic_from_this_build': (traffic_build == lib_sha16()) if traffic_build else None,
          'kernel': (('K1 in fp32 (conv_f32.hip, v_mfma_f32_32x32x2_f32; validation body, un-tuned; an event pair around every call): all %d '
                      if fp32 else
                      'K1 masked conv implicit-GEMM (fwd, dgrad, wgrad; conv_bwd = dgrad + wgrad sharing one launch): all %d ') +
                     'kernel dispatches of %d of the %d timed steps (every %d-th + the mask-update steps), each stamped by its '
                     'own dispatch (hipExtLaunchKernelGGL start/stop events)') % (launches, n_profiled, args.steps, prof_every),
          'profiled_steps': n_profiled,
          'algorithmic_gflop_per_image': flops_per_step / args.batch / 1e9,
          'avg_launch_ms': conv_ms / max(launches, 1),
          'conv_ms_per_step': conv_ms / n_fwd_bwd,
          'by_kind_ms_per_step': {k: prof[k][0] / n_fwd_bwd for k in prof},
          'conv_share_of_step': (conv_ms / n_fwd_bwd) / ms_per_step,
      }
      # the MFMA peak "re-measured on the box before use" -- register-only chains of
      # v_mfma_f32_32x32x16_bf16 (rigl_probe_mfma_bf16), i.e. the clock-and-power-limited dense bf16 rate of THIS box;
      # `peak` (the graded denominator) stays the 2.5 PFLOP/s spec figure of MI355X_MICROARCH.md
      try:
        peak_here = ops.mfma_peak_probe(dev)
        out['roofline']['peak_measured_on_this_box'] = peak_here        # (the bf16 probe; fp32 MFMA runs at 1/16 of it)
        out['roofline']['frac_of_measured_peak'] = (achieved / (peak_here / 16.0 if fp32 else peak_here)) if peak_here > 0 else None
      except Exception as e:  # pylint: disable=broad-except
        out['roofline']['peak_measured_on_this_box'] = None
        print('mfma peak probe failed: %r' % (e,), file=sys.stderr)
      if args.workload.startswith('resnet50'):
        # what the same three GEMMs per layer could reach: each layer at the better of its MFMA and HBM bounds
        lb, lb_mfma, lb_hbm = shapes.resnet50_layerwise_bound(args.batch)
        out['roofline']['layerwise_bound'] = {
            'ms_per_step': lb * 1e3, 'mfma_only_ms': lb_mfma * 1e3, 'hbm_only_ms': lb_hbm * 1e3,
            'frac_of_bound': lb * 1e3 / (conv_ms / n_fwd_bwd) if conv_ms > 0 else 0.0,
            'note': 'sum over layers and fwd/dgrad/wgrad of max(flops / 2.5 PF, algorithmic bytes / 8 TB/s); '
                    'mfma_only_ms / ms_per_step is the highest `frac` any implementation of these layers can reach'}
        # SURVEY 8(d) "report separately, never mix": the reference's EFFECTIVE flop count of the same sparse model
        # (sparse_utils.get_stats: multiplications + additions of one inference with the zeros skipped -- README.md's
        # 0.42x at ERK 0.8, 0.05x at 0.99) next to the dense-equivalent count the MFMA kernels execute and `achieved` divides by
        try:
          sl = shapes.resnet50_stat_layers()
          sp = args.sparsity if args.sparsity is not None else (0.99 if args.workload == 'resnet50_erk99' else 0.8)
          custom = {'initial_conv': 0.0} if args.workload == 'resnet50_erk99' else {}
          dense_f = sparse_utils.get_stats(sl, 0.0, 'random')[0]
          eff_f, eff_bits, eff_s = sparse_utils.get_stats(sl, sp, 'erdos_renyi_kernel', custom_sparsities=custom)
          out['roofline']['effective_gflop_per_image'] = eff_f / 1e9
          out['roofline']['effective'] = {
              'inference_gflop_per_image': eff_f / 1e9, 'dense_inference_gflop_per_image': dense_f / 1e9,
              'ratio_to_dense': eff_f / dense_f, 'model_size_mb': eff_bits / 8 / 1e6, 'sparsity': eff_s,
              'note': 'sparse_utils.get_stats (mults + adds of one forward, zeros skipped); NOT what `achieved` counts: the '
                      'kernels compute on dense storage, algorithmic_gflop_per_image is the dense-equivalent fwd+dgrad+wgrad'}
        except Exception as e:  # pylint: disable=broad-except
          print('effective flops: %r' % (e,), file=sys.stderr)
      # the HBM-bound kernels of the path, against the 8 TB/s HBM3E peak (algorithmic bytes: SURVEY 8d)
      n_params = sum(v.numel for v in g.trainable_variables())
      n_masked = sum(m.numel for m in g.get_masks())
      k3_ms, k3_n = prof['sgd_momentum']
      k2_ms, k2_n = prof['prune_regrow']
      hbm = {}
      if k3_ms > 0:
        steps_k3 = max(n_profiled - n_profiled_updates, 1)      # update iterations skip the weight update (F9)
        gbs = 20.1 * n_params / (k3_ms / steps_k3 * 1e-3) / 1e9
        hbm['masked_sgd_momentum'] = {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                                      'algorithmic_bytes_per_param': 20.1, 'ms_per_step': k3_ms / steps_k3}
      if k2_ms > 0 and k2_n > 0:
        gbs = 8.25 * n_masked / (k2_ms / k2_n * 1e-3) / 1e9
        hbm['prune_regrow'] = {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                               'algorithmic_bytes_per_weight': 8.25, 'ms_per_update': k2_ms / k2_n, 'updates_timed': k2_n,
                               'masked_weights': n_masked}
      dw_ms, dw_n = prof['depthwise']
      if dw_ms > 0 and work['depthwise_bytes'] > 0:
        gbs = work['depthwise_bytes'] * n_fwd_bwd / (dw_ms * 1e-3) / 1e9
        hbm['depthwise_conv'] = {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                                 'algorithmic_bytes_per_step': work['depthwise_bytes'], 'ms_per_step': dw_ms / n_fwd_bwd,
                                 'launches_per_step': dw_n / n_fwd_bwd,
                                 'note': 'K1d fwd + dgrad + wgrad of the 13 dense depthwise 3x3 convs: one bf16 tensor in, one out per pass'}
      out['roofline']['hbm_kernels'] = hbm
    if world == 1 and not args.no_cpu_baseline:
      try:
        from oracle import resnet_cpu
        out['cpu_baseline'] = resnet_cpu.time_cpu_baseline()
      except Exception as e:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': 'failed: %r' % (e,)}
    os.write(real_stdout, (json.dumps(out) + '\n').encode())
  if sync is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
