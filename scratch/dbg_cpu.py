import sys, ctypes, numpy as np, torch, subprocess
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/p2i-gan-benchmark_amd")
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|Flags' | cut -c1-200; nproc; free -g | head -2",shell=True,capture_output=True,text=True).stdout)
from oracle import p2i_oracle as O
lib=ctypes.CDLL("/root/repo/oracle/_build/libp2i_oracle.so")
g=np.load("/root/repo/tests/golden/idw.npz"); T,H,W=16,32,32
for kind in ["gauge","lattice"]:
    mask=torch.from_numpy(g[kind+"_mask"]); mk=mask.reshape(1,H,W).expand(T,H,W)
    tz,ty,tx,pts=O.mask_points(mk); vals=torch.from_numpy(g[kind+"_vals"])
    ref,sel=O.idw_3d_knn(pts,vals,(T,H,W),return_sel=True)
    gx,gy,gz=[torch.linspace(0,1,n) for n in (W,H,T)]
    out=torch.empty(T*H*W); csel=torch.empty(T*H*W,4,dtype=torch.int32)
    P=lambda t: ctypes.c_void_p(t.data_ptr()); pc=pts.contiguous()
    lib.idw_knn4(P(gx),P(gy),P(gz),T,H,W,P(pc),P(vals),pts.shape[0],ctypes.c_float(0.05),P(out),P(csel))
    sameset=(csel.long().sort(1)[0]==sel.sort(1)[0]).all(1)
    print(kind,"C-chain vs torch-on-this-host set match",sameset.float().mean().item(), "golden match torch", float((ref.numpy()==g[kind+"_out"]).mean()), "golden match C", float((out.reshape(T,H,W).numpy()==g[kind+"_out"]).mean()))
print(torch.__config__.show()[:600])
