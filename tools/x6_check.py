"""x6 (bf16-split) conv engine vs the f32-MFMA kernels vs a float64 CPU reference: error and time per layer shape.
usage: python tools/x6_check.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2i-gan-benchmark_amd"))
import ctypes

import torch
import torch.nn.functional as F

from p2igan_bench import ops, _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
CASES = [  # name, cin, cout, (T,)H,W, k3, s3, p3
    ("L0 64@128", 64, 64, (1, 128, 128), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("L1 128@64", 128, 128, (1, 64, 64), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("L2 256@32", 256, 256, (1, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("L3 512@16", 512, 512, (1, 16, 16), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("up 128->64 1x1", 128, 64, (1, 128, 128), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("in 16->64", 16, 64, (1, 128, 128), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("out 64->16 1x1", 64, 16, (1, 128, 128), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("d2d 64->128 s2", 64, 128, (1, 128, 128), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("d2d 256->256", 256, 256, (1, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("d3d 32->64 s122", 32, 64, (16, 64, 64), (3, 3, 3), (1, 2, 2), (1, 1, 1)),
    ("d3d 128->128 s211", 128, 128, (16, 16, 16), (3, 3, 3), (2, 1, 1), (1, 1, 1)),
]


def plan():
    p = (ctypes.c_int * 6)()
    _hip.load().p2i_conv_last_plan(p)
    return tuple(p)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, cin, cout, sp, k3, s3, p3 in CASES:
    spec = ops.ConvSpec(cin, cout, k3, s3, p3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, *sp, generator=g)
    w = torch.randn(cout, cin, *k3, generator=g) / (cin * k3[0] * k3[1] * k3[2]) ** 0.5
    xg = x.to(dev)
    if sp[0] == 1:
        xg = xg.view(B, cin, sp[1], sp[2])
    wp_f, wp_d = ops.weight_pack(w.reshape(cout, cin, -1).to(dev))
    res = {}
    for eng in ("x6", "f32"):
        ops.CONV_ENGINE = eng
        y = ops.conv_fwd(spec, xg, wp_f)
        pf = plan()
        tf = timeit(lambda: ops.conv_fwd(spec, xg, wp_f))
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        dx = ops.conv_dgrad(spec, dy, wp_d, tuple(xg.shape))
        pd = plan()
        td = timeit(lambda: ops.conv_dgrad(spec, dy, wp_d, tuple(xg.shape)))
        res[eng] = (y, dx, tf, td, pf, pd)
    # float64 reference on a sub-batch (CPU)
    nb = 1
    x64 = x[:nb].double()
    w64 = w.double()
    if sp[0] == 1:
        yr = F.conv2d(x64[:, :, 0], w64[:, :, 0], None, s3[1:], p3[1:])
    else:
        yr = F.conv3d(x64, w64, None, s3, p3)
    flops = 2.0 * B * cout * cin * k3[0] * k3[1] * k3[2] * yr[0, 0].numel()
    line = "%-20s" % name
    for eng in ("x6", "f32"):
        y, dx, tf, td, pf, pd = res[eng]
        e = ((y[:nb].double().cpu().reshape(yr.shape) - yr).abs().max() / yr.abs().max()).item()
        line += " | %s fwd %.1e %6.1f TF plan%s  dgrad %6.1f TF plan%s" % (eng, e, flops / tf / 1e12, pf[:3] + pf[5:], flops / td / 1e12, pd[:3] + pd[5:])
    ed = ((res["x6"][1] - res["f32"][1]).abs().max() / res["f32"][1].abs().max()).item()
    line += " | dgrad x6 vs f32 %.1e" % ed
    print(line, flush=True)
