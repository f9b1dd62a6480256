import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/rigl_amd') else os.getcwd())
from rigl_amd import ops
def timeit(fn, iters, warmup=3):
  for _ in range(warmup): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3
dev='cuda:0'; N=128
for (H, Ci, Co, k) in ((14,256,256,3),(7,512,512,3),(14,1024,256,1),(28,512,256,1),(14,1024,512,1),(7,2048,512,1),(14,256,1024,1),(7,512,2048,1)):
  pad = 1 if k == 3 else 0
  set_bytes = N*H*H*(Ci+Co)*2
  copies = max(2, -(-512*(1<<20)//set_bytes))
  xs=[torch.randn(N,H,H,Ci,device=dev).to(torch.bfloat16) for _ in range(copies)]
  ys=[torch.empty(N,H,H,Co,device=dev,dtype=torch.bfloat16) for _ in range(copies)]
  w=(torch.randn(k*k*Ci*Co,device=dev)*0.05).to(torch.bfloat16)
  turn=[0]
  def nxt():
    turn[0]=(turn[0]+1)%copies; return turn[0]
  res=[]
  for v in (0,1):
    ops.tune_set('igemm_n256', v); ops.tune_set('rowstream', 0 if v else 1)
    d=ops.conv_desc(N,H,H,Ci,Co,k,k,1,pad,pad,H,H)
    res.append(timeit(lambda:(lambda i: ops.conv_fwd(d,xs[i],w,ys[i],stats=True))(nxt()), max(10,copies)))
  fl=2.0*N*H*H*k*k*Ci*Co
  print('%2dx%2d %4d->%4d k%d  default %6.1f us (%4.0f TF)  n256 %6.1f us (%4.0f TF)'%(H,H,Ci,Co,k,res[0],fl/res[0]/1e6,res[1],fl/res[1]/1e6), flush=True)
  del xs, ys
