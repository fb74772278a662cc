"""x6c loop phases (diagnostic build, tools/build_stamp.sh): per-wave median cycle sums.
usage: P2I_HIP_LIB=build/ab/libp2i_hip_stamp.so python tools/stamp_x6c.py [B=8]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _hip.load()
lib.p2i_debug_set_stamp.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8 * 65536, dtype=torch.int64, device="cuda")
for name, C, S in (("l0 64@128", 64, 128), ("l1 128@64", 128, 64), ("l2 256@32", 256, 32)):
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = torch.randn(B, C, S, S, device="cuda")
    wp_f, wp_d = ops.weight_pack(torch.randn(C, C, 9, device="cuda") * 0.05)
    f = lambda: ops.conv_fwd(spec, x, wp_f, act=ops.ACT_RELU)
    f(); f(); torch.cuda.synchronize(); buf.zero_()
    lib.p2i_debug_set_stamp(ctypes.c_void_p(buf.data_ptr())); f(); torch.cuda.synchronize(); lib.p2i_debug_set_stamp(ctypes.c_void_p(0))
    r = buf.view(-1, 8).cpu(); r = r[r[:, 4] > 0]; rt = (r[:, 5] >> 32).double(); r[:, 5] &= 0xffffffff; r = r.double(); print('   loop: memtime cycles %.0f, realtime ticks %.0f -> memtime/realtime = %.2f (x100 MHz if realtime is the 100 MHz RTC)' % (r[:, 4].median(), rt.median(), r[:, 4].median() / rt.median())); med = r.median(0).values; nst = 3 * C // 16
    print(f"{name}: per stage: wait {med[0]/nst:6.0f} bar {med[1]/nst:6.0f} issue_w {med[2]/nst:6.0f} mfma {med[3]/nst:6.0f} load_patch {med[5]/nst:6.0f} split {med[7]/nst:6.0f} | loop/stage {med[4]/nst:6.0f} prologue {med[6]:6.0f}", flush=True)
