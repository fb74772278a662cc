"""Every conv-engine launch shape of one configs[1] train step, timed in isolation (HIP events on the launch stream).
usage: python tools/conv_layers_bench.py [B=8] [iters=10] [which=all|fwd|dgrad|wgrad] [filter substring]
Also the workload for the rocprofv3 --pmc passes (profiles/README.md)."""
import os
import sys
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "all"
filt = sys.argv[4] if len(sys.argv) > 4 else ""
dev = "cuda"
S2, S3 = ops.ConvSpec, ops.ConvSpec
LAYERS = [("G l0 3x3 64", S2(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (64, 1, 128, 128)),
          ("G l1 3x3 128", S2(128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (128, 1, 64, 64)),
          ("G l2 3x3 256", S2(256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (256, 1, 32, 32)),
          ("G l3 3x3 512", S2(512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (512, 1, 16, 16)),
          ("G in 16->64 g4", S2(16, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (16, 1, 128, 128)),
          ("G up 512->256 1x1", S2(512, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)), (512, 1, 16, 16)),      # (round 3: projected at the LOW
          ("G up 256->128 1x1", S2(256, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)), (256, 1, 32, 32)),      # resolution, in front of the
          ("G up 128->64 1x1", S2(128, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)), (128, 1, 64, 64)),        # upsampling)
          ("G out 64->16 1x1", S2(64, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0)), (64, 1, 128, 128)),
          ("D2 16->64 s1", S2(16, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (16, 1, 128, 128)),
          ("D2 64->128 s2", S2(64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (64, 1, 128, 128)),
          ("D2 128->256 s2", S2(128, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (128, 1, 64, 64)),
          ("D2 256->256 s1", S2(256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (256, 1, 32, 32)),
          ("D2 256->1 s1", S2(256, 1, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (256, 1, 32, 32)),
          ("D3 1->32 s122", S3(1, 32, (3, 3, 3), (1, 2, 2), (1, 1, 1)), (1, 16, 128, 128)),
          ("D3 32->64 s122", S3(32, 64, (3, 3, 3), (1, 2, 2), (1, 1, 1)), (32, 16, 64, 64)),
          ("D3 64->128 s122", S3(64, 128, (3, 3, 3), (1, 2, 2), (1, 1, 1)), (64, 16, 32, 32)),
          ("D3 128->128 s211", S3(128, 128, (3, 3, 3), (2, 1, 1), (1, 1, 1)), (128, 16, 16, 16)),
          ("D3 128->1 1x1x1", S3(128, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0)), (128, 8, 16, 16))]


def plan(kind):
    lib = ops._hip.load()
    if kind == "wgrad":
        p = (ctypes.c_int * 4)()
        lib.p2i_wgrad_last_plan(p)
    else:
        p = (ctypes.c_int * 6)()
        lib.p2i_conv_last_plan(p)
    return tuple(p)


for name, spec, (c, t, h, w) in LAYERS:
    if filt and filt not in name:
        continue
    is3d = spec.k[0] > 1 or t > 1
    xs = (B, c, t, h, w) if is3d else (B, c, h, w)
    x = torch.randn(xs, device=dev)
    to, ho, wo = spec.out_dims(t, h, w)
    ys = (B, spec.cout, to, ho, wo) if is3d else (B, spec.cout, ho, wo)
    dy = torch.randn(ys, device=dev)
    wp_f, wp_d = ops.weight_pack(torch.randn(spec.cout, spec.cin, spec.ntaps, device=dev) * 0.05)
    fl = 2.0 * B * spec.cout * spec.cin * spec.ntaps * to * ho * wo
    # (the generator's DO-Conv layers have no bias)
    for kind, fn in (("fwd", lambda: ops.conv_fwd(spec, x, wp_f, act=ops.ACT_RELU)),
                     ("dgrad", lambda: ops.conv_dgrad(spec, dy, wp_d, xs, add=x)),
                     ("wgrad", lambda: ops.conv_wgrad(spec, x, dy, want_bias=not name.startswith("G l")))):
        if which not in ("all", kind):
            continue
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"B={B} {name:22s} {kind:5s} {us:8.1f} us {fl / us / 1e6:7.1f} TF plan={plan(kind)}", flush=True)
