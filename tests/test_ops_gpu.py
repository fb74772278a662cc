"""Per-kernel parity: HIP path (through the C ABI) vs the CPU oracle / torch fp32 on the same
seeded inputs.  Tolerances are relative to max-abs of the expected tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_OP = 2e-5       # single fp32 kernel: summation-order noise only
TOL_WGRAD = 1e-4    # long atomically-combined reductions


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from p2igan_bench import ops as o
    o._hip.load()
    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


CONV_CASES = [
    # name, dims(2/3), B, Cin, Cout, in spatial, k, stride, pad, act, bias, residual
    ("g3x3_64", 2, 2, 64, 64, (16, 16), (3, 3), (1, 1), (1, 1), "relu", False, True),
    ("g3x3_128_s16", 2, 3, 128, 128, (16, 16), (3, 3), (1, 1), (1, 1), "none", False, True),
    ("g3x3_tiny4", 2, 2, 512, 512, (4, 4), (3, 3), (1, 1), (1, 1), "none", False, False),
    ("g3x3_w128", 2, 1, 64, 64, (24, 128), (3, 3), (1, 1), (1, 1), "relu", False, False),
    ("g3x3_128_multi", 2, 3, 128, 128, (32, 48), (3, 3), (1, 1), (1, 1), "none", False, True),   # wgrad_x6: 72 tiles on 64 slices (1 or 2 per workgroup)
    ("g3x3_1024_atomic", 2, 1, 1024, 1024, (4, 16), (3, 3), (1, 1), (1, 1), "none", False, False), # 256 channel blocks: one slice, atomic accumulation
    ("g3x3_odd", 2, 3, 32, 48, (19, 21), (3, 3), (1, 1), (1, 1), "relu", True, True),     # partial tiles in H, W and channels
    ("proj1x1", 2, 2, 128, 64, (16, 16), (1, 1), (1, 1), (0, 0), "relu", True, True),
    ("d2d_s2_odd", 2, 2, 16, 32, (20, 22), (3, 3), (2, 2), (1, 1), "leaky", True, False),
    ("d2d_s2", 2, 2, 64, 128, (32, 32), (3, 3), (2, 2), (1, 1), "leaky", True, False),
    ("d2d_to1", 2, 2, 256, 1, (8, 8), (3, 3), (1, 1), (1, 1), "none", True, False),
    ("dense_in16", 2, 2, 16, 64, (16, 16), (3, 3), (1, 1), (1, 1), "none", False, False),
    ("out_tanh", 2, 2, 64, 16, (16, 16), (1, 1), (1, 1), (0, 0), "tanh", False, False),
    ("d3d_first", 3, 2, 1, 32, (8, 16, 16), (3, 3, 3), (1, 2, 2), (1, 1, 1), "leaky", True, False),
    ("d3d_first_tiles", 3, 2, 1, 32, (8, 40, 136), (3, 3, 3), (1, 2, 2), (1, 1, 1), "leaky", True, False),   # 2 column tiles, partial rows
    ("d3d_first_odd", 3, 1, 1, 20, (5, 17, 22), (3, 3, 3), (1, 2, 2), (1, 1, 1), "leaky", True, False),       # T % 4 != 0, odd H, Cout < 32
    ("d3d_mid", 3, 2, 32, 64, (4, 16, 16), (3, 3, 3), (1, 2, 2), (1, 1, 1), "leaky", True, False),
    ("d3d_tstride", 3, 2, 16, 16, (8, 8, 8), (3, 3, 3), (2, 1, 1), (1, 1, 1), "leaky", True, False),
    ("d3d_tstride16", 3, 2, 32, 48, (6, 16, 16), (3, 3, 3), (2, 1, 1), (1, 1, 1), "leaky", True, False),   # x6c with tap slices along t (stride 2 in t)
    ("d3d_t1_16", 3, 1, 16, 32, (5, 16, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1), "none", True, True),            # x6c with tap slices, stride 1, odd T
    ("d3d_1x1x1", 3, 2, 128, 1, (4, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0), "none", True, False),
    ("d2d_64_bias", 2, 2, 64, 128, (16, 32), (3, 3), (1, 1), (1, 1), "leaky", True, False),                  # wgrad_x6 with the bias gradient
    ("d3d_tstride64", 3, 2, 64, 128, (6, 16, 16), (3, 3, 3), (2, 1, 1), (1, 1, 1), "leaky", True, False),     # wgrad_x6 / x6c with t slices, t stride 2
    ("d3d_t1_64", 3, 1, 64, 64, (5, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1), "none", True, True),             # t stride 1, odd T
    ("d2d_256_to1_wide", 2, 2, 40, 1, (9, 70), (3, 3), (1, 1), (1, 1), "none", True, False),                  # o1_fwd_kernel: three column tiles, odd rows
    ("d2d_to1_narrow_oddh", 2, 2, 40, 1, (9, 20), (3, 3), (1, 1), (1, 1), "none", True, False),               # o1_fwd_kernel: one column tile, odd rows (the last row tile's lower lanes)
    ("d2d_to1_crop100", 2, 1, 256, 1, (25, 25), (3, 3), (1, 1), (1, 1), "none", True, False),                 # a 100 x 100 crop's last 2-D layer
    ("d2d_to1_h1_wide", 2, 1, 16, 1, (1, 40), (3, 3), (1, 1), (1, 1), "none", True, False),                   # a single row, two column tiles
]


O1_CASES = ("d2d_to1", "d2d_256_to1_wide", "d2d_to1_narrow_oddh", "d2d_to1_crop100", "d2d_to1_h1_wide")      # o1_fwd_kernel's


def _act_cpu(y, act):
    return {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.2), "tanh": torch.tanh}[act](y)


@pytest.fixture(params=["f32", "x6c", "x6c81", "x6c41", "x6c_sym"])
def engine(request, ops, monkeypatch):
    """Both convolution engines are held to the same oracle: the f32-MFMA kernels and the bf16-split kernels (conv_x6c.hip: producer /
    consumer kernels by default, "x6c_sym" = the symmetric ones).  The latter only take layers with >= 64 workgroups (200 for a variant to be preferred) by default;
    P2I_X6C_MIN_WG=1 (read per call) sends the small test layers they cover through them too, and P2I_X6C_TILE=<waves><channel tiles>
    pins one tile variant (default: 82 = 64 x 256 where it fills the chip, else 81 = 32 x 256, else 41 = 32 x 128 -- producer / consumer
    kernel only -- before any split-K launch)."""
    old = ops.CONV_ENGINE
    ops.CONV_ENGINE = "f32" if request.param == "f32" else "auto"
    if request.param != "f32":
        monkeypatch.setenv("P2I_X6C_MIN_WG", "1")
        if request.param == "x6c_sym":         # the symmetric kernels (all eight waves stage and multiply) with kernel-row stages: the
            monkeypatch.setenv("P2I_X6C_PC", "0")      # round-2 schedule, kept as the A/B baseline and for K = 16 layers
            monkeypatch.setenv("P2I_X6C_TPS", "3")
        elif request.param != "x6c":
            monkeypatch.setenv("P2I_X6C_TILE", request.param[3:])
    yield request.param
    ops.CONV_ENGINE = old


def _last_plan(ops):
    import ctypes
    plan = (ctypes.c_int * 6)()
    ops._hip.load().p2i_conv_last_plan(plan)
    return tuple(plan)


def test_x6c_engine_is_used(ops, monkeypatch):
    """The bf16-split kernel must really run (not silently fall through) where it claims to: on a layer with >= 200 workgroups
    by default, on small 3x3 stride-1 layers once the threshold is lowered, and never with the engine set to f32."""
    old = ops.CONV_ENGINE
    try:
        ops.CONV_ENGINE = "auto"
        spec = ops.ConvSpec(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        wp_f, wp_d = ops.weight_pack(_rand(64, 64, 9, seed=1, scale=0.05).cuda())
        xl = _rand(4, 64, 128, 128, seed=2).cuda()                      # 256 tiles of 64 x 256: default threshold
        yl = ops.conv_fwd(spec, xl, wp_f)
        assert _last_plan(ops)[5] == 7, _last_plan(ops)
        x = xl[:2, :, :32, :32].contiguous()
        ops.conv_fwd(spec, x, wp_f)
        assert _last_plan(ops)[5] != 7                                  # 8 tiles: f32 engine
        monkeypatch.setenv("P2I_X6C_MIN_WG", "1")
        y = ops.conv_fwd(spec, x, wp_f)
        assert _last_plan(ops)[5] == 7, _last_plan(ops)
        ops.conv_dgrad(spec, y, wp_d, tuple(x.shape))
        assert _last_plan(ops)[5] == 7, _last_plan(ops)
        ops.CONV_ENGINE = "f32"
        y32 = ops.conv_fwd(spec, x, wp_f)
        assert _last_plan(ops)[5] != 7
        yl32 = ops.conv_fwd(spec, xl, wp_f)
        assert _last_plan(ops)[5] != 7
        # the split drops only products below 2^-23 of a term: both engines agree to fp32 rounding
        assert rel_err(y.cpu().numpy(), y32.cpu().numpy()) < 5e-6
        assert rel_err(yl.cpu().numpy(), yl32.cpu().numpy()) < 5e-6
    finally:
        ops.CONV_ENGINE = old


def test_x6c_split_k_matches_f32_engine(ops, monkeypatch):
    """Layers with too few tiles for the chip and a linear epilogue run with the channel chunks split over two workgroups per tile
    that ADD into the zeroed destination (plan[2] == 2): bias / residual ride with split 0, the relu' mask distributes over the sum.
    Two addends => the result does not depend on the order.  A forward with an activation gets it (and the residual, which comes
    after the activation) from a second in-place pass."""
    old = ops.CONV_ENGINE
    try:
        spec = ops.ConvSpec(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        wp_f, wp_d = ops.weight_pack(_rand(64, 64, 9, seed=21, scale=0.05).cuda())
        x = _rand(2, 64, 32, 32, seed=22).cuda()          # 32 x 256 tiles: 8 x 2 = 16 workgroups
        res, bias, mk = _rand(2, 64, 32, 32, seed=23).cuda(), _rand(64, seed=24).cuda(), _rand(2, 64, 32, 32, seed=25).cuda()
        ops.CONV_ENGINE = "f32"
        y32 = ops.conv_fwd(spec, x, wp_f, bias=bias, residual=res)
        dx32 = ops.conv_dgrad(spec, x, wp_d, tuple(x.shape), add=res, mask_y=mk, mask_act=ops.ACT_RELU)
        ops.CONV_ENGINE = "auto"
        monkeypatch.setenv("P2I_X6C_MIN_WG", "17")        # 16 < 17 <= 32: only the split-K launch fills "the chip" ...
        monkeypatch.setenv("P2I_X6P_TN1", "0")            # ... once the 32 x 128 tiles (32 workgroups here) are out of the way
        y = ops.conv_fwd(spec, x, wp_f, bias=bias, residual=res)
        assert _last_plan(ops)[5] == 7 and _last_plan(ops)[2] == 2, _last_plan(ops)
        y_again = ops.conv_fwd(spec, x, wp_f, bias=bias, residual=res)
        dx = ops.conv_dgrad(spec, x, wp_d, tuple(x.shape), add=res, mask_y=mk, mask_act=ops.ACT_RELU)
        assert _last_plan(ops)[5] == 7 and _last_plan(ops)[2] == 2, _last_plan(ops)
        assert rel_err(y.cpu().numpy(), y32.cpu().numpy()) < 5e-6 and rel_err(dx.cpu().numpy(), dx32.cpu().numpy()) < 5e-6
        assert torch.equal(y, y_again)                    # run-to-run reproducible
        ya = ops.conv_fwd(spec, x, wp_f, bias=bias, residual=res, act=ops.ACT_RELU)       # activation: applied by a second pass
        assert _last_plan(ops)[5] == 7 and _last_plan(ops)[2] == 2, _last_plan(ops)
        ops.CONV_ENGINE = "f32"
        ya32 = ops.conv_fwd(spec, x, wp_f, bias=bias, residual=res, act=ops.ACT_RELU)
        assert rel_err(ya.cpu().numpy(), ya32.cpu().numpy()) < 5e-6
    finally:
        ops.CONV_ENGINE = old


def test_x6c_presplit_stack_matches_per_call_split(ops, monkeypatch):
    """Weights split ONCE per stack (ops.X6Stack, p2i_x6_split + p2i_conv_*_x6s) give bit-identical results to the per-call split,
    for every layer of the stack, forward and data gradient."""
    monkeypatch.setenv("P2I_X6C_MIN_WG", "1")
    old = ops.CONV_ENGINE
    try:
        ops.CONV_ENGINE = "auto"
        spec = ops.ConvSpec(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        n = 3
        wf = torch.stack([ops.weight_pack(_rand(64, 64, 9, seed=10 + i, scale=0.05).cuda())[0] for i in range(n)])
        wd = torch.stack([ops.weight_pack(_rand(64, 64, 9, seed=10 + i, scale=0.05).cuda())[1] for i in range(n)])
        x = _rand(2, 64, 32, 32, seed=3).cuda()
        plain = [(ops.conv_fwd(spec, x, wf[i].contiguous()), ops.conv_dgrad(spec, x, wd[i].contiguous(), tuple(x.shape))) for i in range(n)]
        assert _last_plan(ops)[5] == 7
        wfs, wds = [wf[i] for i in range(n)], [wd[i] for i in range(n)]
        ops.X6Stack.attach(wf, wfs)
        ops.X6Stack.attach(wd, wds)
        for i in (2, 0, 1):                                   # any order: the stack is split at the first layer that needs it
            y = ops.conv_fwd(spec, x, wfs[i])
            assert _last_plan(ops)[5] == 7
            dx = ops.conv_dgrad(spec, x, wds[i], tuple(x.shape))
            assert _last_plan(ops)[5] == 7
            assert torch.equal(y, plain[i][0]) and torch.equal(dx, plain[i][1])
        assert wfs[0]._x6s[0].wb is not None and wfs[0]._x6s[0] is wfs[2]._x6s[0]
    finally:
        ops.CONV_ENGINE = old


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(ops, engine, case):
    name, nd, B, Cin, Cout, sp, k, st, pd, act, has_bias, has_res = case
    act_code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY, "tanh": ops.ACT_TANH}[act]
    k3 = (1,) + k if nd == 2 else k
    s3 = (1,) + st if nd == 2 else st
    p3 = (0,) + pd if nd == 2 else pd
    spec = ops.ConvSpec(Cin, Cout, k3, s3, p3)
    x = _rand(B, Cin, *sp, seed=1).requires_grad_(True)
    fan = Cin * int(np.prod(k))
    w = _rand(Cout, Cin, *k, seed=2, scale=1.0 / np.sqrt(fan)).requires_grad_(True)
    bias = _rand(Cout, seed=3, scale=0.1).requires_grad_(True) if has_bias else None
    conv = F.conv2d if nd == 2 else F.conv3d
    pre = conv(x, w, bias, st, pd)
    res = _rand(*pre.shape, seed=4) if has_res else None
    y = _act_cpu(pre, act)
    out = y + res if has_res else y
    gout = _rand(*out.shape, seed=5)
    out.backward(gout)

    dev = "cuda"
    wp_f, wp_d = ops.weight_pack(w.detach().to(dev))
    xg = x.detach().to(dev)
    yg = ops.conv_fwd(spec, xg, wp_f, bias.detach().to(dev) if has_bias else None, res.to(dev) if has_res else None, act_code)
    x6c_layer = engine.startswith("x6c") and nd == 2 and k == (3, 3) and st == (1, 1) and name not in ("g3x3_tiny4", "g3x3_w128", "g3x3_1024_atomic") + O1_CASES
    # (those three: a 256-position tile spans 16 or 4 images / is 2 x 128 + halo -- patches above the 384-pixel limit, f32 engine)
    x6c_layer = x6c_layer or (engine.startswith("x6c") and name in ("d3d_tstride16", "d3d_t1_16", "d3d_tstride64", "d3d_t1_64"))
    if x6c_layer and Cin % 16 == 0:
        assert _last_plan(ops)[5] == 7, (name, _last_plan(ops))
    if name in O1_CASES:
        assert _last_plan(ops)[5] == 3, (name, _last_plan(ops))      # single-output-channel bandwidth kernel (o1_fwd_kernel)
    assert rel_err(yg.cpu().numpy(), out.detach().numpy()) < TOL_OP

    # backward: dy_eff = gout * act'(y).  The HIP path gets the post-activation tensor (before residual).
    y_act = y.detach().to(dev).contiguous() if act != "none" else None
    dx = ops.conv_dgrad(spec, gout.to(dev), wp_d, tuple(x.shape), y_act, act_code)
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < TOL_OP
    # fused epilogue: (dx + add) * relu'(mask)
    addt, mk = _rand(*x.shape, seed=6), _rand(*x.shape, seed=7)
    gin = gout * {"none": 1.0, "relu": (y > 0).float(), "leaky": torch.where(y > 0, 1.0, 0.2), "tanh": 1 - y * y}[act]
    dx2 = ops.conv_dgrad(spec, gin.detach().to(dev).contiguous(), wp_d, tuple(x.shape), add=addt.to(dev), mask_y=mk.to(dev), mask_act=ops.ACT_RELU)
    if x6c_layer and Cout % 16 == 0 and Cin > 1:
        assert _last_plan(ops)[5] == 7, (name, _last_plan(ops))
    if engine.startswith("x6c") and name in ("d2d_s2", "d2d_s2_odd", "d3d_mid"):
        assert _last_plan(ops)[5] == 8, (name, _last_plan(ops))      # stride-(.,2,2) data gradient: four parity classes fused on the split pipe
    assert rel_err(dx2.cpu().numpy(), ((x.grad + addt) * (mk > 0)).numpy()) < TOL_OP
    # prologue-free weight gradient on the pre-masked gradient (what the model code uses)
    dwp2, db2 = ops.conv_wgrad(spec, xg, gin.detach().to(dev).contiguous(), want_bias=has_bias)
    vol3 = nd == 3 and k == (3, 3, 3) and pd == (1, 1, 1) and st[0] <= 2
    if (nd == 2 or vol3) and k[-2:] == (3, 3) and st[-2:] == (1, 1) and Cin % 64 == 0 and Cout % 64 == 0 and sp[-2] % 4 == 0 and sp[-1] % 16 == 0:
        import ctypes
        wplan = (ctypes.c_int * 4)()
        ops._hip.load().p2i_wgrad_last_plan(wplan)
        assert wplan[0] == 3, (name, tuple(wplan))          # the bf16-split weight-gradient kernel (wgrad_x6.hip) took it
    assert rel_err(ops.weight_unpack_grad(dwp2, w.detach().to(dev)).cpu().numpy(), w.grad.numpy()) < TOL_WGRAD
    if has_bias:
        assert rel_err(db2.cpu().numpy(), bias.grad.numpy()) < TOL_WGRAD
    dwp, db = ops.conv_wgrad(spec, xg, gout.to(dev), y_act, act_code, want_bias=has_bias)
    dw = ops.weight_unpack_grad(dwp, w.detach().to(dev))
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < TOL_WGRAD
    if has_bias:
        assert rel_err(db.cpu().numpy(), bias.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("O,I,g,ksz,rep", [(64, 64, 1, 3, 0), (128, 128, 1, 3, 0), (64, 16, 4, 3, 4), (16, 64, 4, 1, 0)])
def test_doconv_fold(ops, O, I, g, ksz, rep):
    from oracle import p2i_oracle as orc
    nt = ksz * ksz
    W = _rand(O, I // g, nt, seed=1, scale=0.2).requires_grad_(True)
    D = _rand(I, 9, 9, seed=2, scale=0.05).requires_grad_(True) if ksz == 3 else None
    Dd = torch.eye(9).reshape(1, 9, 9).repeat(I, 1, 1) if ksz == 3 else None
    dow = orc.doconv_fold(W, D, Dd, O, I, g, ksz)                       # (O, I/g, k, k)
    x = _rand(2, I, 8, 8, seed=3)
    y = F.conv2d(x, dow, None, 1, ksz // 2, 1, g)
    if rep:
        y = y + x.repeat_interleave(rep, dim=1)
    gout = _rand(*y.shape, seed=4)
    y.backward(gout)
    dev = "cuda"
    Wg, Dg, Ddg = W.detach().to(dev), (D.detach().to(dev) if D is not None else None), (Dd.to(dev) if Dd is not None else None)
    wp_f, wp_d = ops.doconv_fold(Wg, Dg, Ddg, O, I, g, ksz, identity_rep=rep)
    spec = ops.ConvSpec(I, O, (1, ksz, ksz), (1, 1, 1), (0, ksz // 2, ksz // 2))
    yg = ops.conv_fwd(spec, x.to(dev), wp_f)
    assert rel_err(yg.cpu().numpy(), y.detach().numpy()) < TOL_OP
    dwp, _ = ops.conv_wgrad(spec, x.to(dev), gout.to(dev))
    dW, dD = ops.doconv_fold_bwd(dwp, Wg, Dg, Ddg, O, I, g, ksz)
    assert rel_err(dW.cpu().numpy(), W.grad.numpy()) < TOL_WGRAD
    if ksz == 3:
        assert rel_err(dD.cpu().numpy(), D.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("shape", [(64, 16, 3, 3), (32, 1, 3, 3, 3), (1, 128, 1, 1, 1), (128, 128, 3, 3, 3)])
def test_spectral_norm(ops, shape):
    from oracle import p2i_oracle as orc
    w = _rand(*shape, seed=1, scale=0.1)
    O, K = shape[0], int(np.prod(shape[1:]))
    u = F.normalize(_rand(O, seed=2), dim=0)
    v = F.normalize(_rand(K, seed=3), dim=0)
    p = {"l.weight_orig": w.clone().requires_grad_(True), "l.weight_u": u.clone(), "l.weight_v": v.clone()}
    wn = orc.spectral_norm_weight(p, "l", True)
    gw = _rand(*shape, seed=4)
    wn.backward(gw)
    dev = "cuda"
    ug, vg, wg = u.to(dev), v.to(dev), w.to(dev)
    sigma = ops.spectral_norm(wg, ug, vg, True)
    assert rel_err(ug.cpu().numpy(), p["l.weight_u"].numpy()) < TOL_OP
    assert rel_err(vg.cpu().numpy(), p["l.weight_v"].numpy()) < TOL_OP
    wflat = w.reshape(O, shape[1], -1)
    wp_f, _ = ops.weight_pack(wflat.to(dev), sigma)
    nt = wflat.shape[2]
    got = wp_f[:, :, :O].permute(2, 1, 0).reshape(shape).cpu()
    assert rel_err(got.numpy(), wn.detach().numpy()) < TOL_OP
    # gradient through weight / sigma
    gflat = gw.reshape(O, shape[1], nt)
    dwp = torch.zeros(nt, shape[1], ops.pad32(O))
    dwp[:, :, :O] = gflat.permute(2, 1, 0)
    dw = ops.weight_unpack_grad(dwp.to(dev), wflat.to(dev), wflat.to(dev), sigma, ug, vg)
    assert rel_err(dw.cpu().numpy(), p["l.weight_orig"].grad.reshape(O, shape[1], nt).numpy()) < 1e-4
    # eval mode: sigma from stored u, v, no update
    u2, v2 = ug.clone(), vg.clone()
    s2 = ops.spectral_norm(wg, u2, v2, False)
    assert torch.equal(u2, ug) and torch.equal(v2, vg)
    assert abs(float(s2) - float(sigma)) < 1e-4 * abs(float(sigma))


@pytest.mark.parametrize("T,frac", [(16, 0.2), (8, 0.2), (32, 0.2), (32, 1.0), (16, 0.004)], ids=["T16", "T8", "T32", "T32_dense", "T16_sparse"])
def test_attention_block(ops, T, frac):
    """frac = share of pixels with a non-zero output gradient (the backward kernel lists those per block, ATTN_LIST = 32 per round:
    0.2 of 240 pixels is two rounds, 1.0 eight, 0.004 blocks without any)."""
    from oracle import p2i_oracle as orc
    B, H, W = 2, 12, 20
    x = _rand(B, T, H, W, seed=1).abs()
    w0 = _rand(T, T, 1, seed=2, scale=0.3).requires_grad_(True)
    b0 = _rand(T, seed=3, scale=0.1).requires_grad_(True)
    w1 = _rand(T, T, 1, seed=4, scale=0.3).requires_grad_(True)
    b1 = _rand(T, seed=5, scale=0.1).requires_grad_(True)
    xs = x.permute(0, 2, 3, 1).contiguous().view(B * H * W, T, 1)
    o = orc.attention_block(orc.attention_block(xs, w0, b0), w1, b1).view(B, H, W, T).permute(0, 3, 1, 2)
    gout = _rand(B, T, H, W, seed=6) * (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(7)) < frac)
    o.backward(gout)
    dev = "cuda"
    args = [t.detach().to(dev).contiguous() for t in (x, w0, b0, w1, b1)]
    og = ops.attn_fwd(*args)
    assert rel_err(og.cpu().numpy(), o.detach().numpy()) < TOL_OP
    g = ops.attn_bwd(*args, gout.to(dev).contiguous())
    for got, ref in zip(g, (w0, b0, w1, b1)):
        assert rel_err(got.cpu().numpy(), ref.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("kind", ["gauge", "few", "lattice", "block4"])
def test_idw_matches_oracle_and_golden(ops, kind, golden):
    from oracle import p2i_oracle as orc
    from p2igan_bench.utils import seeded
    T, H, W = 16, 32, 32
    g = golden("idw.npz")
    if kind == "block4":
        mask = seeded.block_mask(H, W, 4, seed=9)
        vals_pts = None
    else:
        mask = torch.from_numpy(g[kind + "_mask"])
    mk = mask.reshape(1, 1, H, W).expand(2, T, H, W).contiguous()
    src = _rand(2, T, H, W, seed=3).abs().requires_grad_(True)
    outs = []
    for b in range(2):
        tz, ty, tx, pts = orc.mask_points(mk[b])
        outs.append(orc.idw_3d_knn(pts, src[b][tz, ty, tx], (T, H, W)))
    ref = torch.stack(outs)
    gout = _rand(2, T, H, W, seed=4)
    ref.backward(gout)
    dev = "cuda"
    og, saved = ops.idw_fwd(src.detach().to(dev), mk.to(dev))
    e = (og.cpu() - ref.detach()).abs()
    # the HIP kernel and the pinned C restatement must select the IDENTICAL 4 points for every voxel
    frac_bad = float((e > 1e-5 * ref.detach().abs().max()).float().mean())
    assert frac_bad == 0.0, f"{kind}: {frac_bad}"
    dv = ops.idw_bwd(gout.to(dev), saved)
    assert rel_err(dv.cpu().numpy(), src.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("kind,shape", [("gauge", (2, 16, 64, 64)), ("gauge", (1, 16, 40, 96)), ("block", (2, 16, 64, 64)), ("pergauge", (2, 8, 48, 48)),
                                        ("gauge", (1, 8, 12, 20)), ("one_empty", (3, 16, 32, 32))])
def test_idw_two_pass_search_equals_index_order_scan(ops, monkeypatch, kind, shape):
    """ops.idw_fwd's default (outward search + replay of the tied voxels, p2i_idw_fwd_ws) against the single-kernel replay of the
    reference's scan (P2I_IDW_FAST=0, the kernel the oracle / golden tests pinned in rounds 1-2): the same four points for every voxel
    -- compared as sets: ties among the four may come out in another order -- and outputs equal to summation order."""
    from p2igan_bench.utils import seeded
    B, T, H, W = shape
    if kind in ("gauge", "one_empty"):       # one mask for every frame: the (t-1)/(t+1) ties the replay pass exists for
        mk = seeded.gauge_mask(H, W, 40 if H * W >= 1024 else 12).reshape(1, 1, H, W).expand(B, T, H, W).contiguous()
        if kind == "one_empty":              # a sample without points between two with: zeros, and nothing of it in the replay list
            mk[1] = 0
    elif kind == "block":
        mk = seeded.block_mask(H, W, 8, seed=5).reshape(1, 1, H, W).expand(B, T, H, W).contiguous()
    else:                                    # a different mask per frame and sample, some frames empty
        g = torch.Generator().manual_seed(11)
        mk = (torch.rand(B, T, H, W, generator=g) < 0.004).float()
        mk[:, 2] = 0
        mk[0, 5] = 0
    src = _rand(B, T, H, W, seed=3).abs().cuda()
    mk = mk.cuda()
    amb = []
    out, (pt_pos, pt_count, sel, selw) = ops.idw_fwd(src, mk, _amb_out=amb)
    counts, replayed = ops.idw_amb_counts(amb[0], B, T * H * W)
    assert bool((replayed <= counts).all())
    if kind in ("gauge", "block") and H * W >= 1024:
        assert int(replayed.max()) * 4 < int(counts.max()), (counts, replayed)   # the fixed-bound phase decides most listed voxels
    # round 3's replay pass (every listed voxel through the reference's heap scan) against round 4's decision from the points within
    # the 4th distance: the same selection sets
    monkeypatch.setenv("P2I_IDW_REPLAY_ALL", "1")
    out_r, (_, _, sel_r, _) = ops.idw_fwd(src, mk)
    monkeypatch.delenv("P2I_IDW_REPLAY_ALL")
    assert torch.equal(sel.view(-1, 4).sort(dim=1).values, sel_r.view(-1, 4).sort(dim=1).values)
    assert float((out - out_r).abs().max()) <= 4e-7 * float(out_r.abs().max())
    monkeypatch.setenv("P2I_IDW_FAST", "0")
    out0, (pt_pos0, pt_count0, sel0, selw0) = ops.idw_fwd(src, mk)
    assert torch.equal(pt_pos[: int(pt_count[0])], pt_pos0[: int(pt_count0[0])]) and torch.equal(pt_count, pt_count0)
    a, b = sel.view(-1, 4).sort(dim=1).values, sel0.view(-1, 4).sort(dim=1).values
    assert torch.equal(a, b), f"{int((a != b).any(dim=1).sum())} voxels select other points"
    assert float((out - out0).abs().max()) <= 4e-7 * float(out0.abs().max())
    assert float((selw.view(-1, 4).sort(dim=1).values - selw0.view(-1, 4).sort(dim=1).values).abs().max()) <= 1e-6
    if kind == "gauge" and H * W >= 1024:
        assert int(counts.min()) > 0                             # the replay pass had voxels to decide
    if kind == "one_empty":
        assert int(counts[1]) == 0 and float(out[1].abs().max()) == 0.0
    assert int(counts.max()) < T * H * W // 4, counts


@pytest.mark.parametrize("H,block,bwd", [(96, 4, True), (128, 4, False)], ids=["96_block4_N9216", "128_block4_N16384"])
def test_idw_dense_mask_matches_oracle(ops, H, block, bwd):
    """The densest masks the loader can draw (`sti` block 4, sti_dataset.py:37-62: one point per 4 x 4 cell, the same cells in every
    frame): N = 9 216 (96 x 96) and 16 384 (128 x 128) points per sample.  These reach what the gauge-mask tests never do: MODE 1's
    LDS window reloaded many times per workgroup (frames t-2 .. t+2 alone hold 2 880 / 5 120 points against a 1 024-point window)
    and idw_bwd_kernel's global-atomic branch (N > 8 192 points do not fit its LDS accumulator).  0 mismatching voxels against the
    pinned C restatement (layer.py:259-293), the backward against autograd of the oracle."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.utils import seeded
    T, W = 16, H
    mk = seeded.block_mask(H, W, block, seed=3).reshape(1, 1, H, W).expand(1, T, H, W).contiguous()
    assert int(mk[0].sum()) == T * (H // block) * (W // block)
    src = _rand(1, T, H, W, seed=3).abs().requires_grad_(bwd)
    tz, ty, tx, pts = orc.mask_points(mk[0])
    ref = orc.idw_3d_knn(pts, src[0][tz, ty, tx], (T, H, W)).unsqueeze(0)
    amb = []
    og, saved = ops.idw_fwd(src.detach().cuda(), mk.cuda(), _amb_out=amb)
    assert int(saved[1][0]) == pts.shape[0]
    e = (og.cpu() - ref.detach()).abs()
    bad = int((e > 1e-5 * float(ref.detach().abs().max())).sum())
    assert bad == 0, f"{bad} voxels differ from the oracle"
    listed, replayed = ops.idw_amb_counts(amb[0], 1, T * H * W)
    assert 0 < int(listed[0]) < T * H * W // 2 and int(replayed[0]) <= int(listed[0])      # pass 2 had rank-4/5 ties to decide
    if bwd:
        gout = _rand(1, T, H, W, seed=4)
        ref.backward(gout)
        dv = ops.idw_bwd(gout.cuda(), saved)
        assert rel_err(dv.cpu().numpy(), src.grad.numpy()) < TOL_WGRAD


def test_idw_fewer_than_four_points_gives_zeros_or_raises(ops, monkeypatch):
    """0 < N < 4 points in a sample: the reference raises in torch.topk(k=4) (layer.py:282, "selected index k out of range").  The
    HIP path cannot raise without a device-to-host sync in the middle of the step; its documented behaviour (include/p2i_hip.h,
    p2i_idw_fwd) is zeros for that sample -- like the empty mask of layer.py:330-332 -- with the other samples unaffected, a zero
    gradient, and the reference's error with P2I_IDW_STRICT=1 (one host sync per call)."""
    from p2igan_bench.utils import seeded
    B, T, H, W = 3, 16, 16, 16
    mk = seeded.gauge_mask(H, W, 12).reshape(1, 1, H, W).expand(B, T, H, W).contiguous()
    mk[1] = 0
    mk[1, 3, 5, 7] = 1
    mk[1, 9, 2, 2] = 1
    mk[1, 9, 11, 4] = 1                                              # sample 1: three points in all
    src = _rand(B, T, H, W, seed=3).abs().cuda()
    out, saved = ops.idw_fwd(src, mk.cuda())
    assert int(saved[1][1]) == 3
    assert float(out[1].abs().max()) == 0.0 and float(out[0].abs().max()) > 0 and float(out[2].abs().max()) > 0
    single, _ = ops.idw_fwd(src[2:].contiguous(), mk[2:].contiguous().cuda())
    assert torch.equal(single[0], out[2])
    dv = ops.idw_bwd(torch.ones_like(out), saved)
    assert float(dv[1].abs().max()) == 0.0 and float(dv[0].abs().max()) > 0
    monkeypatch.setenv("P2I_IDW_STRICT", "1")
    with pytest.raises(RuntimeError, match="out of range"):
        ops.idw_fwd(src, mk.cuda())
    ops.idw_fwd(src[::2].contiguous(), mk[::2].contiguous().cuda())     # samples with >= 4 points (or none) pass the strict check


def test_idw_empty_mask(ops):
    src = _rand(1, 16, 8, 8, seed=1).cuda()
    out, _ = ops.idw_fwd(src, torch.zeros_like(src))
    assert float(out.abs().max()) == 0.0


def test_pooldup(ops):
    from oracle import p2i_oracle as orc
    x = _rand(2, 64, 16, 24, seed=1).requires_grad_(True)
    y = orc.pool_dup(x, 16)
    gout = _rand(*y.shape, seed=2)
    y.backward(gout)
    yg = ops.pooldup_fwd(x.detach().cuda())
    assert torch.equal(yg.cpu(), y.detach())
    dx = ops.pooldup_bwd(x.detach().cuda(), gout.cuda())
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 1e-6


@pytest.mark.parametrize("S", [2, 4, 16, 64])
def test_upmod(ops, S):
    x = _rand(2, 8, S, S, seed=1).requires_grad_(True)
    pos = _rand(1, 1, 2 * S, 2 * S, seed=2, scale=0.5).requires_grad_(True)
    u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    u = u + u * (2 * torch.sigmoid(pos) - 1)
    gout = _rand(*u.shape, seed=3)
    u.backward(gout)
    ug = ops.upmod_fwd(x.detach().cuda(), pos.detach().cuda())
    assert rel_err(ug.cpu().numpy(), u.detach().numpy()) < TOL_OP
    dx, dpos = ops.upmod_bwd(x.detach().cuda(), pos.detach().cuda(), gout.cuda())
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < TOL_OP
    assert rel_err(dpos.cpu().numpy(), pos.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("cin,cout,S", [(64, 32, 8), (128, 64, 16), (32, 16, 4)])
def test_uppos_projection_at_low_resolution_equals_reference_order(ops, cin, cout, S):
    """UPPos (layer.py:384-399) in the reference's order -- upsample, modulate, 1x1 conv + bias, ReLU -- against the build's order:
    1x1 conv at the LOW resolution, then upsample * modulation + bias, ReLU in one kernel (the projection commutes with the two
    per-pixel operators); forward and every gradient (input, weight, bias, pos) through the same op sequence the model code runs."""
    x = _rand(2, cin, S, S, seed=1).requires_grad_(True)
    pos = _rand(1, 1, 2 * S, 2 * S, seed=2, scale=0.5).requires_grad_(True)
    w = _rand(cout, cin, 1, 1, seed=3, scale=1.0 / np.sqrt(cin)).requires_grad_(True)
    b = _rand(cout, seed=4, scale=0.1).requires_grad_(True)
    u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    u = u + u * (2 * torch.sigmoid(pos) - 1)
    r = F.relu(F.conv2d(u, w, b))
    gout = _rand(*r.shape, seed=5)
    r.backward(gout)

    dev = "cuda"
    spec = ops.ConvSpec(cin, cout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    wp_f, wp_d = ops.weight_pack(w.detach().reshape(cout, cin, 1).to(dev))
    xg, pg, bg = x.detach().to(dev), pos.detach().to(dev), b.detach().to(dev)
    v = ops.conv_fwd(spec, xg, wp_f)
    rg = ops.upmod_fwd(v, pg, bias=bg, act=ops.ACT_RELU)
    assert rel_err(rg.cpu().numpy(), r.detach().numpy()) < TOL_OP
    dz, db = ops.act_bwd_bias(gout.to(dev), rg, ops.ACT_RELU)
    assert torch.equal(dz, ops.act_bwd(gout.to(dev), rg, ops.ACT_RELU))
    dv, dpos = ops.upmod_bwd(v, pg, dz)
    dwp, _ = ops.conv_wgrad(spec, xg, dv)
    dw = ops.weight_unpack_grad(dwp, w.detach().reshape(cout, cin, 1).to(dev))
    dx = ops.conv_dgrad(spec, dv, wp_d, tuple(x.shape))
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < TOL_OP
    assert rel_err(dw.cpu().numpy().reshape(w.shape), w.grad.numpy()) < TOL_WGRAD
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < TOL_WGRAD
    assert rel_err(dpos.cpu().numpy(), pos.grad.numpy()) < TOL_WGRAD


@pytest.mark.parametrize("L,win,step,batch", [(40, 16, 4, 32), (21, 16, 4, 3), (7, 16, 4, 32), (16, 16, 4, 2), (33, 8, 8, 32), (19, 16, 1, 5)])
def test_sliding_window_inference_kernels_match_torch(ops, L, win, step, batch):
    """p2i_window_gather / p2i_window_mean (the HIP path of inference.infer_event) against the torch formulation of infer.py:188-262
    (index tensors, index_add_, the last frame repeated, repeated copies not counted), with a stand-in generator; ragged ends, events
    shorter than a window, several batches of windows."""
    from p2igan_bench.inference import infer_event
    H = W = 8
    masked, masks = _rand(1, L, 1, H, W, seed=1).abs(), (_rand(1, L, 1, H, W, seed=2) > 0).float()
    gen = lambda a, b: a * 0.5 + b * 0.25 + (a * b).roll(1, dims=1) - 0.3            # mixes frames of a window: position in the window matters
    ref = infer_event(gen, masked, masks, stride=win, overlap=win - step, output_scale=255.0, max_windows_per_batch=batch)      # CPU: torch path
    got = infer_event(gen, masked.cuda(), masks.cuda(), stride=win, overlap=win - step, output_scale=255.0, max_windows_per_batch=batch)
    assert got.shape == ref.shape
    assert rel_err(got.cpu().numpy(), ref.numpy()) < 1e-6
    nwin = len(range(0, L, step))
    wa, wb = ops.window_gather(masked[0].cuda(), masks[0].cuda(), L, 0, nwin, win, step)
    idx = (torch.arange(win).unsqueeze(0) + torch.arange(0, L, step).unsqueeze(1)).clamp(max=L - 1)
    assert torch.equal(wa.cpu(), masked[0][idx]) and torch.equal(wb.cpu(), masks[0][idx])


@pytest.mark.parametrize("dims", [(8, 8, 2, 4, 4), (32, 32, 8, 16, 16), (8, 8, 4, 8, 8)])
def test_dtail(ops, dims):
    H2, W2, T3, H3, W3 = dims
    o2 = _rand(2, 1, H2, W2, seed=1).requires_grad_(True)
    o3 = _rand(2, 1, T3, H3, W3, seed=2).requires_grad_(True)
    alpha = torch.tensor(0.3, requires_grad=True)
    z2 = o3.mean(dim=2)
    if z2.shape[-2:] != o2.shape[-2:]:
        z2 = F.interpolate(z2, size=o2.shape[-2:], mode="bilinear", align_corners=False)
    fused = (torch.sigmoid(alpha) * o2 + z2).view(2, -1)
    gout = _rand(*fused.shape, seed=3)
    fused.backward(gout)
    fg = ops.dtail_fwd(o2.detach().cuda(), o3.detach().cuda(), alpha.detach().cuda().reshape(1))
    assert rel_err(fg.cpu().numpy(), fused.detach().numpy()) < TOL_OP
    d2, d3, da = ops.dtail_bwd(o2.detach().cuda(), tuple(o3.shape), alpha.detach().cuda().reshape(1), gout.cuda())
    assert rel_err(d2.cpu().numpy(), o2.grad.numpy()) < TOL_OP
    assert rel_err(d3.cpu().numpy(), o3.grad.numpy()) < TOL_OP
    assert abs(float(da) - float(alpha.grad)) < 1e-4 * abs(float(alpha.grad)) + 1e-7


def test_recloss_and_gan_losses(ops):
    from oracle import p2i_oracle as orc
    B, T, H, W = 2, 16, 16, 24
    true = torch.rand(B, T, 1, H, W, generator=torch.Generator().manual_seed(1))
    pred = (torch.tanh(_rand(B, T, 1, H, W, seed=2))).requires_grad_(True)
    loss, pool, reg = orc.reconstruction_loss(pred, true, 0.05)
    loss.backward()
    out3, dpred = ops.recloss(pred.detach().cuda(), true.cuda(), 0.05)
    o = out3.cpu().numpy()
    assert abs(o[0] - float(pool)) < 1e-5 * abs(float(pool))
    assert abs(o[1] - float(reg)) < 1e-4 * abs(float(reg))
    assert abs(o[2] - float(loss)) < 1e-4 * abs(float(loss))
    assert rel_err(dpred.cpu().numpy(), pred.grad.numpy()) < 1e-4
    for lt in ("hinge", "lsgan"):
        a = _rand(2, 64, seed=3).requires_grad_(True)
        b = _rand(2, 64, seed=4).requires_grad_(True)
        ld = (orc.gan_loss(a, True, lt, True) + orc.gan_loss(b, False, lt, True)) * 0.5
        ld.backward()
        l, da, db = ops.gan_loss_d(a.detach().cuda(), b.detach().cuda(), lt)
        assert abs(float(l) - float(ld)) < 1e-5 * abs(float(ld))
        assert rel_err(da.cpu().numpy(), a.grad.numpy()) < 1e-5 and rel_err(db.cpu().numpy(), b.grad.numpy()) < 1e-5
        c = _rand(2, 64, seed=5).requires_grad_(True)
        lg = orc.gan_loss(c, True, lt, False) * 0.01
        lg.backward()
        l2, dc = ops.gan_loss_g(c.detach().cuda(), 0.01, lt)
        assert abs(float(l2) - float(lg)) < 1e-5 * abs(float(lg)) + 1e-9
        assert rel_err(dc.cpu().numpy(), c.grad.numpy()) < 1e-5


def test_adam_matches_torch(ops):
    p = _rand(5000, seed=1)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4, betas=(0.0, 0.99))
    pg, m, v = p.cuda(), torch.zeros(5000).cuda(), torch.zeros(5000).cuda()
    for step in range(1, 4):
        g = _rand(5000, seed=10 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(pg, g.cuda(), m, v, 1e-4, 0.0, 0.99, 1e-8, step)
    assert rel_err(pg.cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert float((pg.cpu() - p).abs().max()) > 1e-5


@pytest.mark.parametrize("mshape", ["hw", "thw", "bthw"])
def test_assemble_batch_bit_exact(ops, mshape):
    """Loader post-processing on the device == numpy's (uint8 -> float32)/255, video*mask (sti_dataset.py:209,223-224), bit for bit."""
    B, T, H, W = 3, 16, 24, 40
    g = torch.Generator().manual_seed(5)
    fr = torch.randint(0, 256, (B, T, H, W), generator=g, dtype=torch.uint8)
    mk = (torch.rand({"hw": (H, W), "thw": (T, H, W), "bthw": (B, T, H, W)}[mshape], generator=g) > 0.9).to(torch.uint8)
    frames, masked, masks = ops.assemble_batch(fr.cuda(), mk.cuda())
    ef = torch.from_numpy(fr.numpy().astype(np.float32) / 255.0).reshape(B, T, 1, H, W)
    em = mk.float().expand(B, T, H, W).reshape(B, T, 1, H, W)
    assert frames.shape == (B, T, 1, H, W)
    assert torch.equal(frames.cpu(), ef) and torch.equal(masks.cpu(), em) and torch.equal(masked.cpu(), ef * em)
    with pytest.raises(RuntimeError):
        ops.assemble_batch(fr.cuda(), mk.cuda()[..., :-1].contiguous())


def test_metric_suite_matches_reference_golden(ops, golden):
    """metrics/metric.py on the HIP path vs the reference's own classes (metrics_32.npz) and the CPU oracle.  The
    contingency counts depend on powf rounding at the thresholds: the GPU's powf and the reference's CPU pow differ in
    the last ulp, so a handful of the 131 072 voxels may land on the other side (bound: 8 per cell, documented)."""
    from oracle import metrics_oracle as mo
    from p2igan_bench.metrics import MetricConfig, RainfallMetricSuite
    from p2igan_bench.utils import seeded
    g = golden("metrics_32.npz")
    suite = RainfallMetricSuite(MetricConfig()).to("cuda")
    for seed in (11, 12):
        p, t = seeded.metric_fields(seed)
        suite.update(p.cuda(), t.cuda())
    table = suite.categorical.counts.cpu().view(4, 4).numpy()
    assert np.abs(table - g["table"].astype(np.int64)).max() <= 8
    assert table.sum(axis=1).tolist() == [2 * 2 * 16 * 32 * 32] * 4
    out = suite.compute()
    vals = dict(zip([str(k) for k in g["keys"]], g["values"]))
    assert sorted(out.keys()) == sorted(vals.keys())
    for k, v in vals.items():
        assert abs(out[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, out[k], v)
    # second instance, FSS alone on one batch, against the oracle restatement
    from p2igan_bench.metrics.metric import FractionalSkillScoreMetric
    p, t = seeded.metric_fields(21, n=1, t=4, h=20, w=28)
    f = FractionalSkillScoreMetric((0.5, 4.0), (1, 2, 8)).to("cuda")
    f.update(p.cuda(), t.cuda())
    ref = mo.fss(p, t, (0.5, 4.0), (1, 2, 8))
    assert np.allclose(f.score_sum.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)
    suite.reset()
    assert suite.compute()["mae"] == 0.0


def test_wgrad_slices_are_bit_reproducible_and_match_atomics(ops):
    """p2i_conv_wgrad_ws: partial tiles stored and summed by wgrad_reduce_kernel -> the same bits on every run (the
    atomic path only agrees to summation-order noise), for a split-heavy layer (64 channels: 256 slices) and a deep one."""
    for (B, C, S) in [(4, 64, 64), (2, 256, 16)]:
        spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        x, dy = _rand(B, C, S, S, seed=1).cuda(), _rand(B, C, S, S, seed=2).cuda()
        old = ops.WGRAD_SLICES
        try:
            ops.WGRAD_SLICES = True
            a, _ = ops.conv_wgrad(spec, x, dy)
            b, _ = ops.conv_wgrad(spec, x, dy)
            ops.WGRAD_SLICES = False
            c, _ = ops.conv_wgrad(spec, x, dy)
        finally:
            ops.WGRAD_SLICES = old
        assert torch.equal(a, b)
        assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1e-5


def test_spectral_norm_batched_matches_per_layer(ops):
    """p2i_spectral_norm_batched (all layers in 4 launches) == p2i_spectral_norm layer by layer, training and eval."""
    shapes = [(64, 16, 3, 3), (128, 64, 3, 3), (32, 1, 3, 3, 3), (1, 128, 1, 1, 1), (256, 256, 3, 3)]
    ws = [_rand(*sh, seed=10 + i, scale=0.1).cuda() for i, sh in enumerate(shapes)]
    for training in (True, False):
        us = [F.normalize(_rand(w.shape[0], seed=20 + i), dim=0).cuda() for i, w in enumerate(ws)]
        vs = [F.normalize(_rand(w.numel() // w.shape[0], seed=30 + i), dim=0).cuda() for i, w in enumerate(ws)]
        u1, v1 = [u.clone() for u in us], [v.clone() for v in vs]
        s1 = [ops.spectral_norm(w, u, v, training) for w, u, v in zip(ws, u1, v1)]
        sb = ops.spectral_norm_batched(ws, us, vs, training)
        for i in range(len(ws)):
            assert abs(float(sb[i]) - float(s1[i])) <= 1e-5 * abs(float(s1[i])), (i, training)
            assert rel_err(us[i].cpu().numpy(), u1[i].cpu().numpy()) < 1e-5 and rel_err(vs[i].cpu().numpy(), v1[i].cpu().numpy()) < 1e-5


def test_doconv_fold_batched_matches_per_layer(ops):
    """p2i_doconv_fold_{fwd,bwd}_batched (a level's same-shape layers in one / two launches) == the per-layer calls."""
    O = I = 64
    layers = []
    for i in range(5):
        W = _rand(O, I, 9, seed=40 + i, scale=0.2).cuda()
        D = _rand(I, 9, 9, seed=50 + i, scale=0.05).cuda()
        Dd = torch.eye(9).reshape(1, 9, 9).repeat(I, 1, 1).cuda()
        layers.append((W, D, Dd))
    one = [ops.doconv_fold(W, D, Dd, O, I, 1, 3) for W, D, Dd in layers]
    bat = ops.doconv_fold_batched(layers, O, I, need_d=True)
    for (f1, d1), (f2, d2) in zip(one, bat):
        assert torch.equal(f1, f2) and torch.equal(d1, d2)
    dwps = [_rand(9, I, O, seed=60 + i).cuda() for i in range(5)]
    ref = [ops.doconv_fold_bwd(g, W, D, Dd, O, I, 1, 3) for g, (W, D, Dd) in zip(dwps, layers)]
    dWs, dDs = ops.doconv_fold_bwd_batched(dwps, layers, O, I)
    for (rW, rD), dW, dD in zip(ref, dWs, dDs):
        assert torch.equal(rW, dW) and rel_err(dD.cpu().numpy(), rD.cpu().numpy()) < 1e-6


def test_weight_pack_unpack_batched_match_per_layer(ops):
    """p2i_weight_pack_batched / p2i_weight_unpack_grad_batched (ten differently shaped layers per launch) == per-layer calls."""
    shapes = [(64, 16, 9), (128, 64, 9), (1, 256, 9), (32, 1, 27), (1, 128, 1)]
    ws = [_rand(*sh, seed=70 + i, scale=0.1).cuda() for i, sh in enumerate(shapes)]
    sig = [torch.tensor([1.5 + 0.1 * i]).cuda() for i in range(len(ws))]
    one = [ops.weight_pack(w, s_) for w, s_ in zip(ws, sig)]
    bat = ops.weight_pack_batched(ws, sig, need_d=True)
    for (f1, d1), (f2, d2) in zip(one, bat):
        assert torch.equal(f1, f2) and torch.equal(d1, d2)
    dwps = [_rand(*f.shape, seed=80 + i).cuda() for i, (f, _) in enumerate(one)]
    us = [F.normalize(_rand(w.shape[0], seed=90 + i), dim=0).cuda() for i, w in enumerate(ws)]
    vs = [F.normalize(_rand(w.shape[1] * w.shape[2], seed=95 + i), dim=0).cuda() for i, w in enumerate(ws)]
    ref = [ops.weight_unpack_grad(g, w, w, s_, u, v) for g, w, s_, u, v in zip(dwps, ws, sig, us, vs)]
    got = ops.weight_unpack_grad_batched(dwps, ws, ws, sig, us, vs)
    for r, g_ in zip(ref, got):
        assert rel_err(g_.cpu().numpy(), r.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("C,S", [(64, 128), (128, 64), (256, 32), (512, 16)], ids=["L0_64@128", "L1_128@64", "L2_256@32", "L3_512@16"])
def test_bf16_split_kernels_hold_fp32_accuracy_vs_float64(ops, monkeypatch, C, S):
    """The "dtype f32" claim of the bf16-split kernels (conv_x6c.hip forward / data gradient, wgrad_x6.hip weight gradient) against a
    FLOAT64 truth, on the four generator level shapes: widening the arithmetic pipe from one f32 MFMA to six bf16 MFMA products must
    not cost accuracy.  Bound: error relative to max-abs < 5e-6 and at most 2x the f32-MFMA engine's own error on the same data."""
    import ctypes
    B = 2
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    g = torch.Generator().manual_seed(100 + C)
    x = torch.randn(B, C, S, S, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    dy = torch.randn(B, C, S, S, generator=g)
    x64, w64, dy64 = x.double().requires_grad_(True), w.double().requires_grad_(True), dy.double()
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(dy64)
    xg, dyg = x.cuda(), dy.cuda()
    wp_f, wp_d = ops.weight_pack(w.reshape(C, C, 9).cuda())
    wplan = (ctypes.c_int * 4)()
    err = {}
    old = ops.CONV_ENGINE
    try:
        for eng in ("f32", "auto"):
            ops.CONV_ENGINE = eng
            if eng == "f32":
                monkeypatch.setenv("P2I_WGRAD_X6", "0")
            else:
                monkeypatch.delenv("P2I_WGRAD_X6", raising=False)
                monkeypatch.setenv("P2I_X6C_MIN_WG", "1")
            y = ops.conv_fwd(spec, xg, wp_f)
            assert (_last_plan(ops)[5] == 7) == (eng == "auto"), (eng, _last_plan(ops))
            dx = ops.conv_dgrad(spec, dyg, wp_d, tuple(xg.shape))
            assert (_last_plan(ops)[5] == 7) == (eng == "auto"), (eng, _last_plan(ops))
            dwp, _ = ops.conv_wgrad(spec, xg, dyg)
            ops._hip.load().p2i_wgrad_last_plan(wplan)
            assert (wplan[0] == 3) == (eng == "auto"), (eng, tuple(wplan))
            dw = ops.weight_unpack_grad(dwp, w.reshape(C, C, 9).cuda()).reshape(C, C, 3, 3)
            err[eng] = (rel_err(y.double().cpu().numpy(), y64.detach().numpy()), rel_err(dx.double().cpu().numpy(), x64.grad.numpy()),
                        rel_err(dw.double().cpu().numpy(), w64.grad.numpy()))
    finally:
        ops.CONV_ENGINE = old
    for i, kind in enumerate(("fwd", "dgrad", "wgrad")):
        assert err["auto"][i] < 5e-6, (kind, err)
        assert err["auto"][i] <= 2.0 * err["f32"][i] + 2e-7, (kind, err)


def test_x6c_fused_strided_dgrad_matches_f32_engine(ops, monkeypatch):
    """Data gradient of stride-(.,2,2) 3x3 / 3x3x3 layers on the bf16-split pipe (patch_gemm_x6c_kernel<8,1,true>: the four input-parity
    classes of the destination in one workgroup, float2 pair stores) against the f32 fused kernel, with the epilogue's add + leaky
    mask; also the split-K form (class pairs ADDED into the zeroed destination) and run-to-run reproducibility."""
    old = ops.CONV_ENGINE
    try:
        for (cin, cout, sp, k3, s3, p3) in [(64, 128, (32, 32), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                            (32, 64, (4, 24, 40), (3, 3, 3), (1, 2, 2), (1, 1, 1))]:
            spec = ops.ConvSpec(cin, cout, k3, s3, p3)
            B = 2
            xs = (B, cin, *sp)
            to, ho, wo = spec.out_dims(*((1,) + sp if len(sp) == 2 else sp))
            ys = (B, cout, ho, wo) if len(sp) == 2 else (B, cout, to, ho, wo)
            _, wp_d = ops.weight_pack(_rand(cout, cin, spec.ntaps, seed=31, scale=0.05).cuda())
            dy, add, mk = _rand(*ys, seed=32).cuda(), _rand(*xs, seed=33).cuda(), _rand(*xs, seed=34).cuda()
            ops.CONV_ENGINE = "f32"
            ref = ops.conv_dgrad(spec, dy, wp_d, xs, add=add, mask_y=mk, mask_act=ops.ACT_LEAKY)
            assert _last_plan(ops)[5] > 10                               # the f32 fused kernel
            ops.CONV_ENGINE = "auto"
            monkeypatch.setenv("P2I_X6C_MIN_WG", "1")
            got = ops.conv_dgrad(spec, dy, wp_d, xs, add=add, mask_y=mk, mask_act=ops.ACT_LEAKY)
            assert _last_plan(ops)[5] == 8 and _last_plan(ops)[2] == 1, _last_plan(ops)
            assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 5e-6
            ntiles = None
            monkeypatch.setenv("P2I_X6C_FUSED_KSPLIT", "1")              # (off by default: slower than the f32 fused kernel where it would apply)
            for mw in (3, 5, 9, 17, 33):                                 # first threshold that only the split-K launch reaches
                monkeypatch.setenv("P2I_X6C_MIN_WG", str(mw))
                g2 = ops.conv_dgrad(spec, dy, wp_d, xs, add=add, mask_y=mk, mask_act=ops.ACT_LEAKY)
                if _last_plan(ops)[5] == 8 and _last_plan(ops)[2] == 2:
                    ntiles = mw
                    break
            assert ntiles is not None, "no split-K launch seen"
            assert rel_err(g2.cpu().numpy(), ref.cpu().numpy()) < 5e-6
            g3 = ops.conv_dgrad(spec, dy, wp_d, xs, add=add, mask_y=mk, mask_act=ops.ACT_LEAKY)
            assert torch.equal(g2, g3)                                   # two addends per element: order-independent
            monkeypatch.delenv("P2I_X6C_FUSED_KSPLIT")
    finally:
        ops.CONV_ENGINE = old


def test_wgrad_x6_producer_consumer_variant_matches_symmetric_kernel(ops, monkeypatch):
    """wgrad_x6p_kernel (P2I_WGRAD_X6_PC=1: consumers with all nine taps, producers staging; measured slower, not the default) computes
    what wgrad_x6_kernel computes (same split and products, another order of the tile rows), incl. bias and t slices."""
    import ctypes
    for (cin, cout, sp, k3, s3, p3, bias) in [(64, 128, (16, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1), True),
                                               (128, 64, (6, 16, 16), (3, 3, 3), (2, 1, 1), (1, 1, 1), True),
                                               (64, 64, (32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1), False)]:
        spec = ops.ConvSpec(cin, cout, k3, s3, p3)
        xs = (3, cin, *sp)
        to, ho, wo = spec.out_dims(*((1,) + sp if len(sp) == 2 else sp))
        ys = (3, cout, ho, wo) if len(sp) == 2 else (3, cout, to, ho, wo)
        x, dy = _rand(*xs, seed=41).cuda(), _rand(*ys, seed=42).cuda()
        res = []
        for pc in ("0", "1"):
            monkeypatch.setenv("P2I_WGRAD_X6_PC", pc)
            dwp, db = ops.conv_wgrad(spec, x, dy, want_bias=bias)
            wplan = (ctypes.c_int * 4)()
            ops._hip.load().p2i_wgrad_last_plan(wplan)
            assert wplan[0] == 3, tuple(wplan)
            res.append((dwp, db))
        # (not bit for bit: the symmetric kernel sums the tile rows of taps 5-8 in another order and the centre tap in two halves)
        assert rel_err(res[1][0].cpu().numpy(), res[0][0].cpu().numpy()) < 2e-6
        if bias:
            assert rel_err(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) < 1e-5      # (bias sums are atomic adds in both)


@pytest.mark.parametrize("C,S", [(256, 32), (512, 16)], ids=["L2_256@32", "L3_512@16"])
def test_presplit_source_planes_are_bit_equal(ops, monkeypatch, C, S):
    """Round-4 experiment (p2i_x6_split_planes / p2i_x6_next_source_planes): a split-pipe convolution that reads its source as
    pre-split bf16 planes runs the same three planes through the same MFMAs as the in-kernel split: forward and data gradient are
    bit-equal on the 32-channel tiles it applies to (32 x 256 at 256 channels, 32 x 128 at 512)."""
    B = 8
    lib = ops._hip.load()
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = _rand(B, C, S, S, seed=1).cuda()
    wp_f, wp_d = ops.weight_pack((_rand(C, C, 9, seed=2) * 0.05).cuda())
    ops.x6_presplit([wp_f, wp_d], [True, True])
    planes = torch.empty(3 * x.numel(), device="cuda", dtype=torch.int16)
    ops._hip.check(lib.p2i_x6_split_planes(x.data_ptr(), planes.data_ptr(), B, C, S * S, torch.cuda.current_stream().cuda_stream), "split")
    res = _rand(B, C, S, S, seed=3).cuda()
    y0 = ops.conv_fwd(spec, x, wp_f, residual=res, act=ops.ACT_RELU)
    assert _last_plan(ops)[5] == 7 and _last_plan(ops)[0] == 32, _last_plan(ops)
    lib.p2i_x6_next_source_planes(planes.data_ptr())
    y1 = ops.conv_fwd(spec, x, wp_f, residual=res, act=ops.ACT_RELU)
    assert torch.equal(y0, y1)
    d0 = ops.conv_dgrad(spec, x, wp_d, tuple(x.shape), add=res, mask_y=res, mask_act=ops.ACT_RELU)
    lib.p2i_x6_next_source_planes(planes.data_ptr())
    d1 = ops.conv_dgrad(spec, x, wp_d, tuple(x.shape), add=res, mask_y=res, mask_act=ops.ACT_RELU)
    assert torch.equal(d0, d1)
    y2 = ops.conv_fwd(spec, x, wp_f, residual=res, act=ops.ACT_RELU)          # the planes were consumed: back to the in-kernel split
    assert torch.equal(y0, y2)
