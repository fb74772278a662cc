import sys, ctypes, numpy as np, torch
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/p2i-gan-benchmark_amd")
from oracle import p2i_oracle as O
from p2igan_bench import ops, _hip
lib=_hip.load()
g=np.load("/root/repo/tests/golden/idw.npz"); T,H,W=16,32,32
kind="gauge"
mask=torch.from_numpy(g[kind+"_mask"]); mk=mask.reshape(1,1,H,W).expand(1,T,H,W).contiguous()
tz,ty,tx,pts=O.mask_points(mk[0])
src=torch.rand(1,T,H,W).cuda(); mkc=mk.cuda()
B=1;Q=T*H*W;dev="cuda"
gx,gy,gz=ops._grid_tables(T,H,W,torch.device("cuda"))
out=torch.empty_like(src); pt_pos=torch.empty(B*Q,device=dev,dtype=torch.int32); pt_count=torch.empty(B,device=dev,dtype=torch.int32)
fc=torch.empty(B*T,device=dev,dtype=torch.int32); xyzn=torch.empty(B*Q*4,device=dev); si=torch.empty(B*Q*4,device=dev,dtype=torch.int32); sw=torch.empty(B*Q*4,device=dev)
p=lambda t:t.data_ptr()
rc=lib.p2i_idw_fwd(p(src),p(mkc),p(gx),p(gy),p(gz),p(out),p(pt_pos),p(pt_count),p(fc),p(xyzn),p(si),p(sw),B,T,H,W,0.05,torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
N=int(pt_count[0]); X=xyzn.view(-1,4)[:N].cpu()
n2=pts.pow(2).sum(-1)
print("px eq",(X[:,0]==pts[:,0]).float().mean().item(),"py",(X[:,1]==pts[:,1]).float().mean().item(),"pz",(X[:,2]==pts[:,2]).float().mean().item(),"pn",(X[:,3]==n2).float().mean().item())
print("grid eq", torch.equal(gx.cpu(),torch.linspace(0,1,W)), torch.equal(gz.cpu(),torch.linspace(0,1,T)))
# compute d2 on GPU with torch ops emulating chain? use CPU C oracle style via float64 emulate for q=3328, idx 20,100
q=3328; t=q//(H*W); y=(q%(H*W))//W; x=q%W
gxc,gyc,gzc=[torch.linspace(0,1,n) for n in (W,H,T)]
import math
def chain(j):
    qx,qy,qz=gxc[x].item(),gyc[y].item(),gzc[t].item()
    f=np.float32
    a0,a1,a2=f(-2)*f(qx),f(-2)*f(qy),f(-2)*f(qz)
    n1=f(f(f(qx)*f(qx)+f(qy)*f(qy))+f(qz)*f(qz))
    px,py,pz=[f(v) for v in pts[j].tolist()]; pn=f(n2[j].item())
    acc=f(a0*px); acc=f(np.float64(a1)*np.float64(py)+np.float64(acc)); acc=f(np.float64(a2)*np.float64(pz)+np.float64(acc)); acc=f(acc+n1); acc=f(acc+pn)
    return acc
print("d2 cpu-emul", [float(chain(j)) for j in (20,100)], [float(np.sqrt(max(chain(j),0))) for j in (20,100)])
d=torch.cdist(O.grid_points(T,H,W)[q:q+1],pts)[0]
print("torch d", d[20].item(), d[100].item(), "d^2", (d[20]**2).item())
