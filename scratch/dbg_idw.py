import sys, ctypes, numpy as np, torch
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/p2i-gan-benchmark_amd")
from oracle import p2i_oracle as O
from p2igan_bench import ops
g=np.load("/root/repo/tests/golden/idw.npz"); T,H,W=16,32,32
for kind in ["gauge","lattice"]:
    mask=torch.from_numpy(g[kind+"_mask"]); mk=mask.reshape(1,1,H,W).expand(1,T,H,W).contiguous()
    tz,ty,tx,pts=O.mask_points(mk[0]); 
    src=torch.rand(1,T,H,W)
    vals=src[0][tz,ty,tx]
    ref,sel=O.idw_3d_knn(pts,vals,(T,H,W),return_sel=True)
    og,saved=ops.idw_fwd(src.cuda(),mk.cuda())
    pt_pos,pt_count,sel_idx,sel_w=saved
    N=int(pt_count[0]); print(kind,"N",N,pts.shape[0])
    pos_ref=(tz*H*W+ty*W+tx).int()
    print("pos order match", torch.equal(pt_pos[:N].cpu(),pos_ref))
    gsel=sel_idx.view(-1,4).cpu().long()   # positions
    rsel=pos_ref.long()[sel]
    bad=(gsel.sort(1)[0]!=rsel.sort(1)[0]).any(1).nonzero().flatten()
    print("bad",bad.numel())
    gp=O.grid_points(T,H,W)
    xyzn=None
    for q in bad[:6].tolist():
        d=torch.cdist(gp[q:q+1],pts)[0]
        dk,ik=torch.topk(d,6,largest=False)
        print(q,"torch top6 d",dk.tolist(),"idx",ik.tolist())
        inv={int(p):i for i,p in enumerate(pos_ref.tolist())}
        print("   gpu idx",[inv[int(p)] for p in gsel[q].tolist()],"ref idx",sel[q].tolist())
