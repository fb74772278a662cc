"""IDW forward alone at configs[1] size (B x 16 x 128 x 128, 79 gauges/frame, and an sti block-10 mask)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import ops
from p2igan_bench.utils import seeded
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, m in (("gauge79", seeded.gauge_mask(128, 128, 79)), ("block10", seeded.block_mask(128, 128, 10))):
    f, k, mk = [t.cuda() for t in seeded.synthetic_batch(B, 16, 128, 128, m)]
    x, mm = k.reshape(B, 16, 128, 128).contiguous(), mk.reshape(B, 16, 128, 128).contiguous()
    for _ in range(2):
        out, _ = ops.idw_fwd(x, mm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        out, _ = ops.idw_fwd(x, mm)
    e1.record(); torch.cuda.synchronize()
    amb = []
    ops.idw_fwd(x, mm, _amb_out=amb)
    c1, c2 = ops.idw_amb_counts(amb[0], B, 16 * 128 * 128)
    und = "  listed for pass 2 / left to the exact replay, sample 0: %d / %d of %d voxels" % (int(c1[0]), int(c2[0]), 16 * 128 * 128)
    print(f"B={B} {name}: {e0.elapsed_time(e1) * 100:.1f} us  checksum {float(out.double().sum()):.6f}{und}", flush=True)
