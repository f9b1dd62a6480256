import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops
dev='cuda:0'; N=128
def timeit(fn, iters, warmup=3):
  for _ in range(warmup): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3
for (H, Ci, Co) in ((14, 256, 1024), (28, 128, 512), (56, 64, 256)):
  copies = max(2, -(-768 * (1 << 20) // (2 * N * H * H * (Ci + Co) * 2)))
  xs = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  ys = [torch.empty(N, H, H, Co, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
  w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
  d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
  turn=[0]
  def nxt():
    turn[0] = (turn[0] + 1) % copies; return turn[0]
  t_s = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i], stats=True))(nxt()), max(10, copies))
  t_f = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i]))(nxt()), max(10, copies))
  print('%s %dx%d %d->%d fwd+stats %.1f us  fwd %.1f us' % (os.environ.get('RIGL_HIP_LIB', 'default'), H, H, Ci, Co, t_s, t_f), flush=True)
