import sys, torch, time
sys.path.insert(0,"/root/repo/p2i-gan-benchmark_amd")
from p2igan_bench import ops
B=int(sys.argv[1]) if len(sys.argv)>1 else 8
which=sys.argv[2] if len(sys.argv)>2 else "all"
n=int(sys.argv[3]) if len(sys.argv)>3 else 20
dev="cuda"
def run(C,S,kind):
    spec=ops.ConvSpec(C,C,(1,3,3),(1,1,1),(0,1,1))
    x=torch.randn(B,C,S,S,device=dev); w=torch.randn(C,C,9,device=dev)*0.05
    wp_f,wp_d=ops.weight_pack(w)
    dy=torch.randn(B,C,S,S,device=dev)
    fl=2.0*B*C*C*9*S*S
    if kind=="fwd": f=lambda: ops.conv_fwd(spec,x,wp_f,act=ops.ACT_RELU)
    elif kind=="dgrad": f=lambda: ops.conv_dgrad(spec,dy,wp_d,tuple(x.shape),add=x)
    else: f=lambda: ops.conv_wgrad(spec,x,dy)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/n
    import ctypes
    plan=(ctypes.c_int*6)(); ops._hip.load().p2i_conv_last_plan(plan)
    print(f"B={B} C={C} S={S} {kind}: {us:.1f} us  {fl/us/1e6:.1f} TF plan={tuple(plan)}",flush=True)
for (C,S) in [(64,128),(128,64),(256,32),(512,16)]:
    for kind in ["fwd","dgrad","wgrad"]:
        if which in ("all",kind): run(C,S,kind)
