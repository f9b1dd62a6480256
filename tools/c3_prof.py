import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from rigl_amd import ops
dev='cuda:0'; N,H,C=128,56,64
x=torch.randn(N,H,H,C,device=dev).to(torch.bfloat16); dy=torch.randn(N,H,H,C,device=dev).to(torch.bfloat16)
w=(torch.randn(9*C*C,device=dev)*0.05).to(torch.bfloat16); dw=torch.empty(9*C*C,device=dev)
d=ops.conv_desc(N,H,H,C,C,3,3,1,1,1,H,H)
for _ in range(3): ops.conv_bwd(d,x,dy,w,dw,need_dx=True); ops.conv_fwd(d,x,w,stats=True)
torch.cuda.synchronize(); ops.prof_collect(); ops.prof_enable(True)
for _ in range(5): ops.conv_bwd(d,x,dy,w,dw,need_dx=True); ops.conv_fwd(d,x,w,stats=True)
torch.cuda.synchronize(); ops.prof_enable(False)
rows=ops.prof_collect_launches()
for r in rows[-8:]: print(r[0], '%.1f us'%(r[2]*1e3))
