"""End-to-end parity on the GPU: generator / discriminator / losses / full G+D train steps through
the drop-in API vs (a) golden vectors captured from the genuine reference and (b) the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north_star: 1e-4 relative (to max-abs), fp32

CFG32 = {"model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": 32, "w": 32, "sample_length": 16}},
         "loss": {"use_gan": 1, "gan_loss": "hinge", "k1_weight": 0.05, "adversarial_weight": 0.01},
         "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from p2igan_bench import _hip
    _hip.load()
    return torch.device("cuda:0")


def _batch32():
    from p2igan_bench.utils import seeded
    h = w = 32
    m0 = seeded.gauge_mask(h, w, 20)
    m1 = seeded.block_mask(h, w, 4)
    f0, k0, mk0 = seeded.synthetic_batch(1, 16, h, w, m0, seed=2024)
    f1, k1, mk1 = seeded.synthetic_batch(1, 16, h, w, m1, seed=3024)
    return torch.cat([f0, f1]), torch.cat([k0, k1]), torch.cat([mk0, mk1])


def _build(dev, h=32, w=32):
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    cfg = dict(CFG32, data={"train": {"h": h, "w": w, "sample_length": 16}})
    G = build_generator(cfg).to(dev)
    D = build_discriminator(cfg).to(dev)
    G.load_state_dict(seeded.seeded_generator_state(h, w))
    D.load_state_dict(seeded.seeded_discriminator_state())
    return cfg, G, D


def test_train_steps_match_reference_golden(dev, golden):
    from p2igan_bench.engine import TrainEngine
    g = golden("e2e_32.npz")
    cfg, G, D = _build(dev)
    eng = TrainEngine(G, D, cfg)
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    r = eng.train_step(frames, masked, masks)
    assert rel_err(r["preds"].cpu().numpy(), g["preds"]) < TOL
    assert rel_err(r["logits_fake"].cpu().numpy(), g["logits_fake"]) < TOL
    assert rel_err(r["logits_real"].cpu().numpy(), g["logits_real"]) < TOL
    for k in ("loss_g", "loss_d", "pool", "reg"):
        assert abs(float(r[k]) - float(g[k])) <= TOL * abs(float(g[k])), k
    # adv is evaluated AFTER the discriminator's Adam step (beta1=0: every weight moves by ~lr*sign(g)), so a
    # gradient whose sign flips under fp32 summation-order noise shifts it: looser, documented tolerance
    assert abs(float(r["adv"]) - float(g["adv"])) <= 2e-3 * abs(float(g["adv"]))
    # gradients of step 0 are still in the flat grad buffers
    gparams = dict(G.named_parameters())
    dparams = dict(D.named_parameters())
    for k in g.files:
        if k.startswith("ggradnorm/"):
            n = k.split("/", 1)[1]
            assert abs(float(gparams[n].grad.norm()) - float(g[k])) <= 1e-3 * float(g[k]) + 1e-7, k
        if k.startswith("ggrad/"):
            assert rel_err(gparams[k.split("/", 1)[1]].grad.cpu().numpy(), g[k]) < 1e-3, k
    # D grads were overwritten by nothing after the D step (G step does not touch them)
    for k in g.files:
        if k.startswith("dgradnorm/"):
            n = k.split("/", 1)[1]
            assert abs(float(dparams[n].grad.norm()) - float(g[k])) <= 1e-3 * float(g[k]) + 1e-7, k
        if k.startswith("dgrad/"):
            assert rel_err(dparams[k.split("/", 1)[1]].grad.cpu().numpy(), g[k]) < 1e-3, k
    assert float(dparams["alpha3d"].grad.abs().sum()) == 0.0      # never receives a gradient (p2igan.py:145,170)
    gsd, dsd = G.state_dict(), D.state_dict()
    for k in g.files:
        if k.startswith("g1sum/"):
            assert abs(float(gsd[k[6:]].double().sum()) - float(g[k])) <= 2e-3 * max(1.0, abs(float(g[k]))), k
        if k.startswith("d1sum/"):
            assert abs(float(dsd[k[6:]].double().sum()) - float(g[k])) <= 2e-3 * max(1.0, abs(float(g[k]))), k
    assert rel_err(dsd["d3d.0.weight_u"].cpu().numpy(), g["d1/d3d.0.weight_u"]) < 1e-4
    assert rel_err(dsd["d2d.6.weight_v"].cpu().numpy(), g["d1/d2d.6.weight_v"]) < 1e-4
    r = eng.train_step(frames, masked, masks)
    assert abs(float(r["loss_g"]) - float(g["loss_g_step1"])) < 1e-3 * abs(float(g["loss_g_step1"]))
    r = eng.train_step(frames, masked, masks)
    assert abs(float(r["loss_g"]) - float(g["loss_g_step2"])) < 1e-3 * abs(float(g["loss_g_step2"]))
    assert abs(float(r["loss_d"]) - float(g["loss_d_step2"])) < 1e-3 * abs(float(g["loss_d_step2"]))
    assert rel_err(G.state_dict()["Convsin.0.main.0.W"].cpu().numpy(), g["g3/Convsin.0.main.0.W"]) < 1e-3
    assert rel_err(D.state_dict()["d2d.2.bias"].cpu().numpy(), g["d3/d2d.2.bias"]) < 1e-3


def test_generator_only_training_step_matches_oracle(dev):
    """The reference's non-GAN config (p2igan_baseline.json: use_gan 0): reconstruction loss only, no discriminator.  The engine's
    side-stream preparation must work without a D step; predictions, loss and every generator gradient against the oracle."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.utils import seeded
    cfg, G, _ = _build(dev)
    cfg = dict(cfg, loss=dict(cfg["loss"], use_gan=0))
    eng = TrainEngine(G, None, cfg)
    assert not eng.use_gan and eng.prep_overlap
    frames, masked, masks = _batch32()
    r = eng.train_step(frames.to(dev), masked.to(dev), masks.to(dev))
    ref = orc.TrainState(seeded.seeded_generator_state(32, 32), None, cfg["loss"], cfg["train"]["optimizer"]).step(frames, masked, masks, keep_grads=True)
    assert rel_err(r["preds"].cpu().numpy(), ref["preds"].numpy()) < TOL
    assert abs(float(r["loss_g"]) - ref["loss_g"]) <= TOL * abs(ref["loss_g"])
    gparams = dict(G.named_parameters())
    for n, gr in ref["ggrads"].items():
        if gr is not None:
            assert abs(float(gparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    r2 = eng.train_step(frames.to(dev), masked.to(dev), masks.to(dev))
    assert bool(torch.isfinite(r2["preds"]).all()) and float(r2["loss_g"]) < float(r["loss_g"])


def test_intermediate_taps_match_oracle(dev, golden):
    """Stage by stage against the CPU oracle AND the reference's own taps (e2e_32.npz) on the same batch: IDW output,
    Convsin output, deepest pooled tensor, Decoder[3], both UPPos outputs named in p2igan.py:91-105, and the two
    discriminator branch outputs (localises a regression)."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.utils import seeded
    g = golden("e2e_32.npz")
    cfg, G, D = _build(dev)
    frames, masked, masks = _batch32()
    taps, dtaps, htaps, hdtaps = {}, {}, {}, {}
    with torch.no_grad():
        ref = orc.generator_forward(seeded.seeded_generator_state(32, 32), masked, masks, taps)
        G.debug_taps = htaps
        out = G(masked.to(dev), masks.to(dev))
        G.debug_taps = None
        lref = orc.discriminator_forward(seeded.seeded_discriminator_state(), frames, training=True, taps=dtaps)
        D.train()
        D.debug_taps = hdtaps
        lg = D(frames.to(dev))
        D.debug_taps = None
    assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL
    assert rel_err(lg.cpu().numpy(), lref.numpy()) < TOL
    for k in ("idw", "x_", "x_8", "dec3", "res1", "res3"):
        assert rel_err(htaps[k].cpu().numpy(), taps[k].numpy()) < TOL, k
    for k in ("out2d", "out3d"):
        assert rel_err(hdtaps[k].cpu().numpy(), dtaps[k].numpy()) < TOL, k
    for k in ("idw", "dec3", "res1", "res3"):                       # the genuine reference's forward hooks
        assert rel_err(htaps[k].cpu().numpy(), g["tap_" + k]) < TOL, k
    # the reference's D taps were captured on D(preds.detach()) of step 0: same weights, its own preds
    D2 = _build(dev)[2]
    D2.train()
    D2.debug_taps = hdtaps
    with torch.no_grad():
        D2(torch.from_numpy(g["preds"]).to(dev))
    for k in ("out2d", "out3d"):
        assert rel_err(hdtaps[k].cpu().numpy(), g["tap_" + k + "_fake"]) < TOL, k
    # eval mode does not move u, v
    D.eval()
    u0 = D.d2d[0].weight_u.clone()
    with torch.no_grad():
        D(frames.to(dev))
    assert torch.equal(u0, D.d2d[0].weight_u)


def _hip_full_step(dev, golden, name, h):
    """One TrainEngine step at full spatial size vs the genuine reference's step (tests/fullsize.py rules)."""
    import fullsize
    from p2igan_bench.engine import TrainEngine
    g = golden(name)
    cfg, G, D = _build(dev, h, h)
    eng = TrainEngine(G, D, cfg)
    frames, masked, masks = [t.to(dev) for t in fullsize.batch_for(g, h, h)]
    taps = {}
    G.debug_taps = taps
    r = eng.train_step(frames, masked, masks)
    G.debug_taps = None
    ggrads = {n: p.grad for n, p in G.named_parameters() if p.grad is not None}
    dgrads = {n: p.grad for n, p in D.named_parameters() if p.grad is not None}
    fullsize.check(g, r, ggrads, dgrads, G.state_dict(), D.state_dict(), taps)
    assert float(dgrads["alpha3d"].abs().sum()) == 0.0
    return eng, (frames, masked, masks)


@pytest.fixture(params=["auto", "f32"])
def engines(request, monkeypatch):
    """Both sets of convolution kernels are held to the reference's full-size goldens: "auto" = the default (bf16-split x6c forward /
    dgrad and wgrad_x6 weight gradient where they apply, f32-MFMA kernels elsewhere), "f32" = f32-MFMA kernels only."""
    from p2igan_bench import ops
    old = ops.CONV_ENGINE
    ops.CONV_ENGINE = request.param
    if request.param == "f32":
        monkeypatch.setenv("P2I_WGRAD_X6", "0")
    yield request.param
    ops.CONV_ENGINE = old


def test_full_size_train_step_128_matches_reference_golden(dev, golden, engines):
    """configs[1] geometry (B=2 of the B=8 workload; 128x128; 79-gauge 'stis' mask and an 'sti' block-10 mask): forward,
    backward through D and G, wgrad (256-slice path), spectral-norm gradient, Adam -- vs e2e_128.npz."""
    _hip_full_step(dev, golden, "e2e_128.npz", 128)


def test_full_size_train_step_256_matches_reference_golden(dev, golden, engines):
    """configs[3] geometry (256x256, 316 gauges): the re-sized tiles through a whole train step -- vs e2e_256.npz."""
    _hip_full_step(dev, golden, "e2e_256.npz", 256)


@pytest.mark.parametrize("T,h,w", [(32, 32, 32), (8, 32, 32), (16, 32, 48), (16, 40, 24)], ids=["T32", "T8", "T16_32x48", "T16_40x24"])
def test_long_window_t32_matches_oracle(dev, T, h, w):
    """BASELINE configs[4] (T=32), the third window length the AttentionBlock kernels exist for (T=8), and non-square frames at T=16
    (the reference's own geometry in both dimensions' roles: one full train step against the oracle, incl. the IDW tap).  NO REFERENCE BEHAVIOUR EXISTS for T != 16 (layer.py:310, p2igan.py:46,66,79 raise):
    parity UNPINNED -- this checks the HIP path against the oracle's restatement of the same generalisation (SURVEY.md H5:
    AttentionBlock(T), Convsin T->4T, base_channel 4T, discriminator in_channels T) for one full train step at 32x32."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    cfg = dict(CFG32, data={"train": {"h": h, "w": w, "sample_length": T}})
    gs, ds = seeded.seeded_generator_state(h, w, t=T), seeded.seeded_discriminator_state(t=T)
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    G.load_state_dict(gs)
    D.load_state_dict(ds)
    f0, k0, m0 = seeded.synthetic_batch(1, T, h, w, seeded.gauge_mask(h, w, 20), seed=2024)
    f1, k1, m1 = seeded.synthetic_batch(1, T, h, w, seeded.block_mask(h, w, 4), seed=3024)
    frames, masked, masks = torch.cat([f0, f1]), torch.cat([k0, k1]), torch.cat([m0, m1])
    eng = TrainEngine(G, D, cfg)
    taps = {}
    G.debug_taps = taps
    got = eng.train_step(frames.to(dev), masked.to(dev), masks.to(dev))
    G.debug_taps = None
    st = orc.TrainState(gs, ds, cfg["loss"], cfg["train"]["optimizer"])
    otaps = {}
    ref = st.step(frames, masked, masks, keep_grads=True, taps=otaps)
    assert got["preds"].shape == (2, T, 1, h, w) and got["logits_real"].shape == (2, (h // 4) * (w // 4))
    assert rel_err(taps["idw"].cpu().numpy(), otaps["idw"].detach().numpy()) < 1e-5
    assert rel_err(got["preds"].cpu().numpy(), ref["preds"].numpy()) < TOL
    assert rel_err(got["logits_fake"].cpu().numpy(), ref["logits_fake"].numpy()) < TOL
    assert rel_err(got["logits_real"].cpu().numpy(), ref["logits_real"].numpy()) < TOL
    for k in ("loss_g", "loss_d", "pool", "reg"):
        assert abs(float(got[k]) - ref[k]) <= TOL * abs(ref[k]), k
    gparams, dparams = dict(G.named_parameters()), dict(D.named_parameters())
    for n, gr in ref["ggrads"].items():
        assert abs(float(gparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    for n, gr in ref["dgrads"].items():
        if gr is not None:
            assert abs(float(dparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    # (the generator's gradients pass through the discriminator AFTER its Adam step -- beta1 = 0: a sign update, a gradient whose sign
    # flips under summation-order noise moves a weight by 2e-4 -- hence the documented 2e-3 of fullsize.check for the added shapes;
    # the first two cases keep the 1e-3 they were written with)
    gtol = 1e-3 if (h, w) == (32, 32) else 2e-3
    for n in ("Convsin.0.main.0.W", "Convsin.0.main.0.D", "ConvsOut.0.main.0.W", "input.layers.0.conv.weight"):
        assert rel_err(gparams[n].grad.cpu().numpy(), ref["ggrads"][n].numpy()) < gtol, n


def test_reference_api_gan_loss_matches_oracle(dev):
    """modules.gan_loss (losses.py:232-253): every (real / fake / generator) x (hinge, lsgan) term and its gradient vs the
    oracle; the hinge discriminator single terms run through the saturated other side of the fused D-loss kernel."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.modules import gan_loss
    x = (torch.randn(3, 1024, generator=torch.Generator().manual_seed(5)) * 1.5)
    for lt in ("hinge", "lsgan"):
        for is_real, is_disc in ((True, True), (False, True), (True, False)):
            xr = x.clone().requires_grad_(True)
            ref = orc.gan_loss(xr, is_real, lt, is_disc, 0.9, 0.1)
            ref.backward()
            xg = x.clone().to(dev).requires_grad_(True)
            got = gan_loss(xg, is_real, loss_type=lt, is_disc=is_disc, target_real_label=0.9, target_fake_label=0.1)
            (got * 3.0).backward()
            assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (lt, is_real, is_disc)
            assert rel_err(xg.grad.cpu().numpy(), 3.0 * xr.grad.numpy()) < 1e-5, (lt, is_real, is_disc)
    with pytest.raises(ValueError):
        gan_loss(x.to(dev), True, loss_type="wgan", is_disc=True)


def test_generator_128_matches_reference_golden(dev, golden):
    from p2igan_bench.utils import seeded
    g = golden("g_128.npz")
    cfg, G, _ = _build(dev, 128, 128)
    frames, masked, masks = seeded.synthetic_batch(1, 16, 128, 128, seeded.gauge_mask(128, 128, 79))
    G.eval()
    with torch.no_grad():
        preds = G(masked.to(dev), masks.to(dev))
    assert rel_err(preds.cpu().numpy()[0, :, 0, ::3, ::3], g["preds_s3"]) < TOL
    assert abs(float(preds.double().sum()) - float(g["preds_sum"])) < 1e-4 * float(g["preds_abs_sum"])


def test_empty_mask_and_eval_logits(dev, golden):
    from p2igan_bench.utils import seeded
    g = golden("infer_32.npz")
    cfg, G, D = _build(dev)
    h = w = 32
    ev = seeded.synthetic_event(40, h, w, seed=99).float() / 255.0
    frames = ev.reshape(1, 40, 1, h, w)[:, :16].contiguous()
    G.eval(); D.eval()
    with torch.no_grad():
        z = G(torch.zeros_like(frames).to(dev), torch.zeros_like(frames).to(dev))
        logits = D(frames.to(dev))
    assert rel_err(z.cpu().numpy(), g["empty"]) < TOL
    assert rel_err(logits.cpu().numpy(), g["logits_eval"]) < TOL


def test_full_size_properties(dev):
    """Config A size (B=2,16,128,128): size-independent properties — determinism of the forward,
    batch independence (sample b's output does not depend on its batch-mates), tanh range."""
    from p2igan_bench.utils import seeded
    cfg, G, D = _build(dev, 128, 128)
    m = seeded.gauge_mask(128, 128, 79)
    f, k, mk = seeded.synthetic_batch(2, 16, 128, 128, m)
    G.eval()
    with torch.no_grad():
        a = G(k.to(dev), mk.to(dev))
        b = G(k.to(dev), mk.to(dev))
        c = G(k[1:].to(dev), mk[1:].to(dev))
    assert torch.equal(a, b)
    assert rel_err(c.cpu().numpy(), a[1:].cpu().numpy()) < 1e-5
    assert float(a.abs().max()) <= 1.0


def test_infer_event_matches_reference_golden(dev, golden):
    """Batched sliding-window inference (infer.py:188-245) incl. the last-frame padding branch."""
    from p2igan_bench.inference import infer_event
    from p2igan_bench.utils import seeded
    g = golden("infer_32.npz")
    cfg, G, _ = _build(dev)
    h = w = 32
    m = seeded.gauge_mask(h, w, 24, seed=7)
    L = 40
    ev = seeded.synthetic_event(L, h, w, seed=99).float() / 255.0
    frames = ev.reshape(1, L, 1, h, w)
    masks = m.reshape(1, 1, 1, h, w).expand(1, L, 1, h, w).contiguous()
    G.eval()
    comp = infer_event(G, (frames * masks).to(dev), masks.to(dev))
    assert rel_err(comp.cpu().numpy(), g["comp"]) < TOL


def test_train_and_infer_scripts_run(dev, tmp_path):
    """scripts/train.py + scripts/infer.py end to end on synthetic 32x32 events (config surface of SURVEY.md §5)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd", "scripts"))
    import infer as infer_script
    import train as train_script
    cfg = json.load(open(os.path.join(root, "p2i-gan-benchmark_amd", "p2igan_bench", "config", "p2igan_gan_baseline.json")))
    for k in ("train", "test"):
        cfg["data"][k].update(w=32, h=32)
    cfg["data"]["train"]["data_root"] = "synthetic://4"
    cfg["data"]["valid"]["data_root"] = "synthetic://2"
    cfg["data"]["test"]["mask"]["block_sizes"] = [4]
    cfg["data"]["train"]["mask"]["block_sizes"] = [4]
    cfg["train"].update(batch_size=2, iterations=3, log_step=1)
    cfg["save_dir"] = str(tmp_path / "w")
    cp = tmp_path / "cfg.json"
    json.dump(cfg, open(cp, "w"))
    train_script.main(["--config", str(cp), "--run-validation"])
    ck = torch.load(tmp_path / "w" / "latest.pt", weights_only=True)
    assert set(ck) == {"epoch", "global_step", "generator", "optimizer_g", "discriminator", "optimizer_d"}
    assert ck["global_step"] == 3
    infer_script.main(["--config", str(cp), "--model-dir", str(tmp_path / "w"), "--output", str(tmp_path / "out.zarr")])
    from p2igan_bench.data import zarr_lite
    out = zarr_lite.Group(str(tmp_path / "out.zarr"))
    arr = out["event_01"][:]
    assert arr.shape == (40, 1, 32, 32) and arr.dtype == np.float32 and float(arr.min()) >= 0.0
    assert out.attrs["model_name"] == "p2igan" and out.attrs["output_scale"] == 255.0


def _dp_worker(rank, world, port, out):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from p2igan_bench.engine import TrainEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the one GPU of the test box
    dev = torch.device("cuda:0")
    cfg, G, D = _build(dev)
    eng = TrainEngine(G, D, cfg, distributed=True)
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    sl = slice(rank, rank + 1)
    assert eng.dp_overlap                       # default: per-level buckets launched from inside the generator's backward
    for _ in range(2):
        eng.train_step(frames[sl].contiguous(), masked[sl].contiguous(), masks[sl].contiguous())
    res = {"g": eng.gp.flat.cpu(), "d": eng.dp.flat.cpu()}
    # the bucketed exchange against the single flat all-reduce, one step from the same state (equal up to the float atomics
    # of the few small reductions of a backward pass)
    grads = []
    for overlap in (True, False):
        _, G2, D2 = _build(dev)
        e2 = TrainEngine(G2, D2, cfg, distributed=True)
        e2.dp_overlap = overlap
        e2.train_step(frames[sl].contiguous(), masked[sl].contiguous(), masks[sl].contiguous())
        grads.append(e2.gp.grad.cpu().clone())
    res["bucket_vs_flat"] = float((grads[0] - grads[1]).abs().max() / grads[1].abs().max())
    torch.save(res, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process(dev, tmp_path):
    """TrainEngine's distributed path (broadcast + flat-bucket all-reduce after each backward + fused Adam): two
    ranks with one sample each must reproduce one process at global batch 2.  gloo moves the buckets here
    because both ranks sit on the single GPU of the test box; on a node the same code runs over RCCL."""
    import os
    import torch.multiprocessing as mp
    from p2igan_bench.engine import TrainEngine
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["d"], r1["d"])          # ranks stay bit-identical
    assert r0["bucket_vs_flat"] == 0.0 and r1["bucket_vs_flat"] == 0.0, (r0["bucket_vs_flat"], r1["bucket_vs_flat"])       # bit-identical (round 4)
    cfg, G, D = _build(dev)
    eng = TrainEngine(G, D, cfg)
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    for _ in range(2):
        eng.train_step(frames, masked, masks)
    # step-1 Adam (beta1 = 0) is a sign update: compare with a budget of a few flipped signs per million weights
    for key, flat in (("g", eng.gp.flat), ("d", eng.dp.flat)):
        diff = (flat.cpu() - r0[key]).abs()
        assert float((diff > 5e-5).float().mean()) < 2e-3, key
        assert float(diff.max()) <= 4.1e-4, key


def test_inference_variant_and_weight_cache(dev, golden):
    """P2IGenerator(inference=True) with the converted checkpoint reproduces the reference (eval_variant_32.npz: the
    reference's own inference=True model, which make_golden.py checked to equal its training variant exactly);
    repeated no-grad forwards reuse the packed weights until the parameters change."""
    from p2igan_bench.models.p2igan import P2IGenerator, fold_generator_state_dict
    from p2igan_bench.utils import seeded
    g = golden("eval_variant_32.npz")
    cfg = dict(CFG32, data={"train": {"h": 32, "w": 32, "sample_length": 16}})
    Ge = P2IGenerator(cfg, inference=True).to(dev)
    sd = fold_generator_state_dict(seeded.seeded_generator_state(32, 32))
    Ge.load_state_dict(sd, strict=True)
    Ge.eval()
    m0 = seeded.gauge_mask(32, 32, 20)
    frames, masked, masks = [t.to(dev) for t in seeded.synthetic_batch(1, 16, 32, 32, m0, seed=2024)]
    with torch.no_grad():
        y1 = Ge(masked, masks)
        n_cached = len(Ge._wp_cache)
        vers = [v[0] for v in Ge._wp_cache.values()]
        y2 = Ge(masked, masks)
    assert rel_err(y1.cpu().numpy(), g["preds"]) < TOL
    assert n_cached == 34 + 3 and [v[0] for v in Ge._wp_cache.values()] == vers      # 32 res convs + in + out, 3 UPPos proj: all hits
    assert torch.equal(y1, y2)
    # a torch-level weight change is seen through the version counters ...
    with torch.no_grad():
        Ge.ConvsOut[0].main[0].W.mul_(0.5)
        y3 = Ge(masked, masks)
    assert not torch.allclose(y3, y1)
    # ... and the training variant (which folds D every forward when grads are needed) gives the same frames
    _, G, _ = _build(dev)
    G.eval()
    with torch.no_grad():
        yt = G(masked, masks)
    assert rel_err(yt.cpu().numpy(), g["preds"]) < TOL
    with pytest.raises(RuntimeError):          # forward-only: grads w.r.t. folded kernels are not defined by the reference
        Ge(masked, masks)


def test_two_eager_runs_are_bit_identical_for_five_steps(dev):
    """Run-to-run reproducibility (the reference on CPU is deterministic): two engines from the same state, five full G+D steps each
    -- every loss, the predictions, both flat parameter buffers, both gradient buffers and the spectral-norm vectors are BIT-equal.
    (Rounds 1-3: a few small reductions were float atomics and two runs drifted to 4e-3 in loss_g by step 5.)  Also with the float
    atomics switched back on (P2I_DETERMINISTIC=0 in a fresh process is the A/B: tools/determinism_probe.py)."""
    from p2igan_bench.engine import TrainEngine
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    runs = []
    for _ in range(2):
        cfg, G, D = _build(dev)
        eng = TrainEngine(G, D, cfg)
        outs = []
        for _ in range(5):
            r = eng.train_step(frames, masked, masks)
            outs.append({k: r[k].clone() for k in ("loss_g", "loss_d", "rec", "adv", "pool", "reg", "preds", "logits_real", "logits_fake")})
        runs.append((outs, eng.gp.flat.clone(), eng.dp.flat.clone(), eng.gp.grad.clone(), eng.dp.grad.clone(),
                     {n: b.clone() for n, b in D.named_buffers()}))
    (oa, ga, da, gga, dga, ba), (ob, gb, db, ggb, dgb, bb) = runs
    for step, (a, b) in enumerate(zip(oa, ob)):
        for k in a:
            assert torch.equal(a[k], b[k]), (step, k)
    assert torch.equal(ga, gb) and torch.equal(da, db) and torch.equal(gga, ggb) and torch.equal(dga, dgb)
    for n in ba:
        assert torch.equal(ba[n], bb[n]), n


@pytest.mark.parametrize("mode", ["graph", "tape"])
def test_graph_replay_matches_eager_steps(dev, mode):
    """TrainEngine.capture(): the replay of the whole G+D step (device-side Adam step counter) -- as a hipGraph, or through the
    library's launch tape (p2i_tape_replay, the native step sequencer: every launch, memset and stream dependency of the recorded
    step re-enqueued by one C call) -- follows the eager engine.  Compared at step 2 (one eager warm-up step + one replay): the float atomics of a few small reductions
    make two EAGER runs drift apart too (beta1 = 0 Adam amplifies sign flips: ~1e-7 at step 2, ~4e-3 in loss_g by step
    5, measured), so a later comparison would test that chaos, not the replay."""
    from p2igan_bench.engine import TrainEngine
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    res = []
    for graph in (False, True):
        cfg, G, D = _build(dev)
        eng = TrainEngine(G, D, cfg)
        if graph:
            assert eng.capture(frames, masked, masks, warmup=1, mode=mode) == 1
            assert eng.opt_g.step_count == 1 and int(eng.opt_g.step_dev) == 1
            if mode == "tape":
                nk, nm, ne, nstreams = eng.tape_info()
                assert nk > 300 and nm > 5 and ne >= 20 and nstreams == 3, (nk, nm, ne, nstreams)     # main + the two side streams
        else:
            eng.train_step(frames, masked, masks)
        r = eng.train_step(frames, masked, masks)
        assert eng.opt_g.step_count == 2 and eng.opt_d.step_count == 2
        if graph:
            assert int(eng.opt_g.step_dev) == 2 and int(eng.opt_d.step_dev) == 2
        res.append(({k: float(r[k]) for k in ("loss_g", "loss_d", "rec")}, r["preds"].clone(), eng.gp.flat.clone(), eng.dp.flat.clone()))
        if graph and mode == "tape":                  # a third and fourth step through the tape: the counters and the weights keep moving
            before = eng.gp.flat.clone()
            eng.train_step(frames, masked, masks)
            r4 = eng.train_step(frames, masked, masks)
            assert int(eng.opt_g.step_dev) == 4 and float((eng.gp.flat - before).abs().max()) > 0 and bool(torch.isfinite(r4["loss_g"]))
    (la, pa, ga, da), (lb, pb, gb, db) = res
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-4 * abs(la[k]), (k, la[k], lb[k])
    assert rel_err(pb.cpu().numpy(), pa.cpu().numpy()) < 1e-4
    # (a gradient whose sign flips under summation-order noise moves a weight by up to lr/sqrt(1-beta2) = 1e-3 in one step)
    assert float((ga - gb).abs().max()) <= 2e-3 and float((da - db).abs().max()) <= 2e-3
    assert float((ga - gb).abs().mean()) <= 2e-6 and float((da - db).abs().mean()) <= 2e-6


def test_two_rank_rccl_bench_launch(dev):
    """bench.py --gpus 2 over RCCL (backend 'nccl'): needs two visible GPUs, so it runs on multi-GPU nodes only (the
    one-GPU test box skips it; the launcher itself is covered on CPU by tests/test_bench_launch_cpu.py)."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    # what RCCL itself saw, and the exposed exchange time of a step (bench.py::collective_evidence, TrainEngine.exchange_marks)
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["ranks_seen"] == 2
    assert line["rccl"]["exchange_d_ms_per_step"] > 0 and line["rccl"]["exchange_g_exposed_ms_per_step"] >= 0


def test_autograd_step_matches_direct_step(dev):
    """TrainEngine's direct step (plain forward / backward functions, gradients written in place) and the autograd-driven step
    (the drop-in modules + loss.backward(), what a user of the reference API runs) execute the same kernels: one step from the
    same state must agree to summation-order noise."""
    from p2igan_bench.engine import TrainEngine
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    res = []
    for direct in (True, False):
        cfg, G, D = _build(dev)
        eng = TrainEngine(G, D, cfg)
        assert eng.direct
        eng.direct = direct
        r = eng.train_step(frames, masked, masks)
        res.append(({k: float(r[k]) for k in ("loss_g", "loss_d", "rec", "adv", "pool", "reg")}, r["preds"].clone(),
                    eng.gp.grad.clone(), eng.dp.grad.clone()))
    (la, pa, ga, da), (lb, pb, gb, db) = res
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(la[k]) + 1e-9, (k, la[k], lb[k])
    assert torch.equal(pa, pb)
    assert rel_err(gb.cpu().numpy(), ga.cpu().numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), da.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("gan_type", ["hinge", "lsgan"])
def test_overlapped_step_matches_serial_step(dev, monkeypatch, gan_type):
    """The step's stream-level overlap (round 3: weight preparation, the real half of the D step incl. its backward, the
    reconstruction loss and the generator's fold / unpack kernels on side streams) against the same step issued on ONE stream in the
    reference's order.  Round 4: BIT-identical -- the small reductions are summed in a fixed order now (p2i_det_workspace), and the one
    order that does differ (D's real and fake halves add their weight gradients in the other order) is a sum of two addends."""
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import p2igan as net_fns
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    res = []
    for overlapped in (True, False):
        cfg, G, D = _build(dev)
        cfg = dict(cfg, loss=dict(cfg["loss"], gan_loss=gan_type))
        monkeypatch.setattr(net_fns, "SIDE_WGRAD", overlapped)
        monkeypatch.setattr(net_fns, "LATE_JOIN", overlapped)
        eng = TrainEngine(G, D, cfg)
        assert eng.direct and eng.prep_overlap
        eng.prep_overlap = overlapped
        r = eng.train_step(frames, masked, masks)
        res.append(({k: float(r[k]) for k in ("loss_g", "loss_d", "rec", "adv", "pool", "reg")}, r["preds"].clone(),
                    r["logits_real"].clone(), r["logits_fake"].clone(), eng.gp.grad.clone(), eng.dp.grad.clone()))
        r2 = eng.train_step(frames, masked, masks)                 # (the side streams' buffers are reused: a second step must run clean)
        assert all(bool(torch.isfinite(r2[k]).all()) for k in ("loss_g", "loss_d", "preds"))
    (la, pa, ra, fa, ga, da), (lb, pb, rb, fb, gb, db) = res
    for k in la:
        assert la[k] == lb[k], (k, la[k], lb[k])
    assert torch.equal(pa, pb) and torch.equal(ra, rb) and torch.equal(fa, fb)       # forwards run the same kernels on the same data
    assert torch.equal(ga, gb), float((ga - gb).abs().max())
    assert torch.equal(da, db), float((da - db).abs().max())


def test_nsgan_matches_oracle_and_rejects_out_of_range(dev):
    """'nsgan' = nn.BCELoss on the raw logits (losses.py:201-202): runs only for logits in [0, 1] (torch raises otherwise)."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.modules import gan_loss
    x = torch.rand(2, 1024, generator=torch.Generator().manual_seed(9)) * 0.98 + 0.01
    for is_real, is_disc in ((True, True), (False, True), (True, False)):
        xr = x.clone().requires_grad_(True)
        ref = orc.gan_loss(xr, is_real, "nsgan", is_disc, 0.9, 0.1)
        ref.backward()
        xg = x.clone().to(dev).requires_grad_(True)
        got = gan_loss(xg, is_real, loss_type="nsgan", is_disc=is_disc, target_real_label=0.9, target_fake_label=0.1)
        got.backward()
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (is_real, is_disc)
        assert rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) < 1e-5, (is_real, is_disc)
    with pytest.raises(RuntimeError):
        gan_loss((x + 0.5).to(dev), True, loss_type="nsgan", is_disc=True)


def test_long_window_t32_full_size_properties(dev):
    """BASELINE configs[4] geometry (T=32, 128x128; per-GPU batch 4 -> B=2 here): NO reference behaviour (parity unpinned, see
    test_long_window_t32_matches_oracle for the oracle check at 32x32).  Size-independent properties at full size: the forward is
    deterministic and batch-independent, a train step gives finite losses and moves every trainable tensor except the unused alpha3d."""
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    T, h, w = 32, 128, 128
    cfg = dict(CFG32, data={"train": {"h": h, "w": w, "sample_length": T}})
    torch.manual_seed(7)
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    f, k, m = [t.to(dev) for t in seeded.synthetic_batch(2, T, h, w, seeded.gauge_mask(h, w, 79))]
    G.eval()
    with torch.no_grad():
        a, b, c = G(k, m), G(k, m), G(k[1:], m[1:])
    assert a.shape == (2, T, 1, h, w) and torch.equal(a, b) and float(a.abs().max()) <= 1.0
    assert rel_err(c.cpu().numpy(), a[1:].cpu().numpy()) < 1e-5
    eng = TrainEngine(G, D, cfg)
    before = eng.gp.flat.clone()
    r = eng.train_step(f, k, m)
    for key in ("loss_g", "loss_d", "pool", "reg", "adv"):
        assert torch.isfinite(r[key]).all(), key
    assert r["logits_real"].shape == (2, (h // 4) * (w // 4))
    assert bool(torch.isfinite(eng.gp.flat).all()) and float((eng.gp.flat - before).abs().max()) > 0
    for n, p in G.named_parameters():
        if p.requires_grad:
            assert float(p.grad.abs().max()) > 0, n


def test_hi_res_256_train_step_properties(dev):
    """BASELINE configs[3] geometry (256x256, per-GPU batch 4 -> B=2 here): on top of the reference golden at B=1
    (test_full_size_train_step_256_matches_reference_golden), two engines from the same state take bit-identical steps apart
    from the float-atomic reductions (losses to 1e-6), and the step equals the B=1 steps' sample-wise forward."""
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.utils import seeded
    h = w = 256
    f, k, m = [t.to(dev) for t in seeded.synthetic_batch(2, 16, h, w, seeded.gauge_mask(h, w, 316))]
    outs = []
    for _ in range(2):
        cfg, G, D = _build(dev, h, w)
        r = TrainEngine(G, D, cfg).train_step(f, k, m)
        outs.append(r)
    for key in ("loss_g", "loss_d", "pool", "reg"):
        a, b = float(outs[0][key]), float(outs[1][key])
        assert abs(a - b) <= 1e-6 * abs(a) and a == a, key
    assert torch.equal(outs[0]["preds"], outs[1]["preds"])
    cfg, G, _ = _build(dev, h, w)
    G.eval()
    with torch.no_grad():
        single = G(k[:1], m[:1])
    assert rel_err(single.cpu().numpy(), outs[0]["preds"][:1].cpu().numpy()) < 1e-5


def test_bench_batch_b8_matches_oracle(dev):
    """The bench's OWN regime (bench.py: configs[1], B=8, 128x128, 79 gauges/frame, batch built like bench.py:244-245): one full
    TrainEngine step against the CPU oracle's step on the same inputs.  At B=8 the tile picker takes different kernels than at the
    B=2 of the reference goldens (64 x 256 x6c tiles on the 64/128-channel levels instead of 32 x 256 / split-K / f32), so this is the
    launch regime the headline number is measured in; the recorded kernel keys assert that."""
    from oracle import p2i_oracle as orc
    from p2igan_bench import ops
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    B, T, h, w = 8, 16, 128, 128
    cfg = dict(CFG32, data={"train": {"h": h, "w": w, "sample_length": T}})
    gs, ds = seeded.seeded_generator_state(h, w), seeded.seeded_discriminator_state()
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    G.load_state_dict(gs)
    D.load_state_dict(ds)
    frames, masked, masks = seeded.synthetic_batch(B, T, h, w, seeded.gauge_mask(h, w, 79), seed=2024)
    eng = TrainEngine(G, D, cfg)
    ops.PROFILE = ops.KernelProfile()
    try:
        got = eng.train_step(frames.to(dev), masked.to(dev), masks.to(dev))
        keys = ops.PROFILE.summary()
    finally:
        ops.PROFILE = None
    n82 = sum(v["launches"] for k, v in keys.items() if k.startswith(("patch_gemm_x6c_kernel<8, 2, false", "patch_gemm_x6p_kernel<2, ")))
    n81 = sum(v["launches"] for k, v in keys.items() if k.startswith(("patch_gemm_x6c_kernel<8, 1, false", "patch_gemm_x6p_kernel<1, 9, false")))
    nfu = sum(v["launches"] for k, v in keys.items() if k.startswith(("patch_gemm_x6c_kernel<8, 1, true", "patch_gemm_x6p_kernel<1, 9, true")))
    assert nfu >= 9, sorted(keys)                 # the discriminators' strided data gradients: fused parity classes on the split pipe
    # generator levels 0 and 1: 16 forward + 16 dgrad launches of the 64 x 256 tile (+ the discriminator's 2-D layers it takes)
    assert n82 >= 32 and n81 >= 32, sorted(keys)
    assert any(k.startswith("wgrad_x6_kernel") and v["launches"] >= 32 for k, v in keys.items()), sorted(keys)
    ref = orc.TrainState(gs, ds, cfg["loss"], cfg["train"]["optimizer"]).step(frames, masked, masks, keep_grads=True)
    assert rel_err(got["preds"].cpu().numpy(), ref["preds"].numpy()) < TOL
    assert rel_err(got["logits_fake"].cpu().numpy(), ref["logits_fake"].numpy()) < TOL
    assert rel_err(got["logits_real"].cpu().numpy(), ref["logits_real"].numpy()) < TOL
    for k in ("loss_g", "loss_d", "pool", "reg"):
        assert abs(float(got[k]) - ref[k]) <= TOL * abs(ref[k]), (k, float(got[k]), ref[k])
    assert abs(float(got["adv"]) - ref["adv"]) <= 2e-3 * abs(ref["adv"])           # after D's Adam step: see fullsize.check
    gparams, dparams = dict(G.named_parameters()), dict(D.named_parameters())
    for n, gr in ref["ggrads"].items():
        assert abs(float(gparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    for n, gr in ref["dgrads"].items():
        if gr is not None:
            assert abs(float(dparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    import fullsize
    for n in ("Decoder.0.layers.3.main.1.main.0.W", "Decoder.1.layers.0.main.0.main.0.D", "Decoder.2.layers.1.main.0.main.0.W",
              "UP.0.proj.weight", "Convsin.0.main.0.W", "input.layers.1.conv.weight"):
        assert fullsize.grad_err(gparams[n].grad.cpu().numpy(), ref["ggrads"][n].numpy()) < 1e-3, n
    for n in ("d2d.2.weight_orig", "d3d.4.weight_orig", "d3d.6.bias", "d2d.6.weight_orig"):
        assert fullsize.grad_err(dparams[n].grad.cpu().numpy(), ref["dgrads"][n].numpy()) < 1e-3, n


@pytest.mark.parametrize("T,h,w", [(16, 256, 256), (32, 128, 128)], ids=["configs3_256x256_b4", "configs4_T32_b4"])
def test_per_gpu_batch4_matches_oracle(dev, T, h, w):
    """The per-GPU batch of BASELINE configs[3] (256 x 256, B = 16 over 4 GPUs) and configs[4] (T = 32, B = 32 over 8 GPUs): ONE
    TrainEngine step at B = 4 against the CPU oracle's step on the same inputs.  The tile picker keys on the workgroup count
    (conv_x6c.hip: 200 preferred, 64 the last resort), so B = 4 takes other tiles than the B = 1 / B = 2 of the goldens and property
    tests; the recorded kernel keys assert that the split-pipe forward / data-gradient kernels, the fused strided data gradients
    and wgrad_x6 really ran.  T = 32 has no reference behaviour (layer.py:310): held to the generalised oracle, parity unpinned."""
    from oracle import p2i_oracle as orc
    from p2igan_bench import ops
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    B = 4
    cfg = dict(CFG32, data={"train": {"h": h, "w": w, "sample_length": T}})
    gs, ds = seeded.seeded_generator_state(h, w, t=T), seeded.seeded_discriminator_state(t=T)
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    G.load_state_dict(gs)
    D.load_state_dict(ds)
    # sample 0 carries the config's gauge density (316 at 256 x 256, 79 at 128 x 128), the others block masks (1 024 / 256 points per
    # sample): the oracle's brute-force 4-NN search is the slow part of this test
    ms = [seeded.gauge_mask(h, w, 79 * (h // 128) ** 2)] + [seeded.block_mask(h, w, 32, seed=5 + i) for i in range(B - 1)]
    parts = [seeded.synthetic_batch(1, T, h, w, m, seed=2024 + 1000 * i) for i, m in enumerate(ms)]
    frames, masked, masks = (torch.cat([p[j] for p in parts]) for j in range(3))
    eng = TrainEngine(G, D, cfg)
    ops.PROFILE = ops.KernelProfile()
    try:
        got = eng.train_step(frames.to(dev), masked.to(dev), masks.to(dev))
        keys = ops.PROFILE.summary()
    finally:
        ops.PROFILE = None
    nx6 = sum(v["launches"] for k, v in keys.items() if k.startswith(("patch_gemm_x6c_kernel<8, ", "patch_gemm_x6p_kernel<")) and "strided dgrad" not in k)
    nfu = sum(v["launches"] for k, v in keys.items() if "strided dgrad, 4 parity classes per workgroup" in k and "x6" in k)
    nwg = sum(v["launches"] for k, v in keys.items() if k.startswith("wgrad_x6_kernel"))
    assert nx6 >= 64 and nfu >= 9 and nwg >= 32, sorted(keys)         # the generator's 32 + 32 3x3 launches, D's strided dgrads, 32 weight gradients
    ref = orc.TrainState(gs, ds, cfg["loss"], cfg["train"]["optimizer"]).step(frames, masked, masks, keep_grads=True)
    assert rel_err(got["preds"].cpu().numpy(), ref["preds"].numpy()) < TOL
    assert rel_err(got["logits_fake"].cpu().numpy(), ref["logits_fake"].numpy()) < TOL
    assert rel_err(got["logits_real"].cpu().numpy(), ref["logits_real"].numpy()) < TOL
    for k in ("loss_g", "loss_d", "pool", "reg"):
        assert abs(float(got[k]) - ref[k]) <= TOL * abs(ref[k]), (k, float(got[k]), ref[k])
    assert abs(float(got["adv"]) - ref["adv"]) <= 2e-3 * abs(ref["adv"])           # after D's Adam step: see fullsize.check
    gparams, dparams = dict(G.named_parameters()), dict(D.named_parameters())
    for n, gr in ref["ggrads"].items():
        assert abs(float(gparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    for n, gr in ref["dgrads"].items():
        if gr is not None:
            assert abs(float(dparams[n].grad.norm()) - float(gr.norm())) <= 1e-3 * float(gr.norm()) + 1e-7, n
    import fullsize
    for n in ("Decoder.0.layers.3.main.1.main.0.W", "Decoder.2.layers.1.main.0.main.0.W", "Decoder.3.layers.0.main.1.main.0.D",
              "UP.1.proj.weight", "Convsin.0.main.0.W", "input.layers.1.conv.weight"):
        assert fullsize.grad_err(gparams[n].grad.cpu().numpy(), ref["ggrads"][n].numpy()) < 1e-3, n
    for n in ("d2d.2.weight_orig", "d3d.4.weight_orig", "d3d.6.bias", "d2d.6.weight_orig"):
        assert fullsize.grad_err(dparams[n].grad.cpu().numpy(), ref["dgrads"][n].numpy()) < 1e-3, n


ZCFG = {"seed": 11, "model": {"name": "p2igan", "in_channels": 1},
        "loss": {"use_gan": 1, "gan_loss": "hinge", "k1_weight": 0.05, "adversarial_weight": 0.01},
        "train": {"batch_size": 8, "num_workers": 0, "device_assemble": True, "optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}


def _zarr_cfg(root):
    return dict(ZCFG, data={"train": {"data_root": str(root), "w": 128, "h": 128, "sample_length": 16,
                                      "mask": {"type": "sti", "block_sizes": [10]}}})


def _zarr_rank_batches(cfg, rank, world, nsteps):
    """What scripts/train.py feeds rank `rank`: seed_everything(seed, rank) (mask draws from numpy's, crop offsets from python's
    generator), P2IDataModule(cfg, rank, world) on the windowed store, epoch 1 of the sharded sampler, uint8 hand-over."""
    import random
    from p2igan_bench.data.dataloader import P2IDataModule
    random.seed(cfg["seed"] + rank)
    np.random.seed(cfg["seed"] + rank)
    loader = P2IDataModule(cfg, rank, world).train_dataloader()
    loader.sampler.set_epoch(1)
    out = []
    for batch in loader:
        assert len(batch) == 2 and batch[0].dtype == torch.uint8 and tuple(batch[0].shape) == (8, 16, 128, 128)
        out.append(batch)
        if len(out) == nsteps:
            break
    return out


def _zarr_dp_worker(rank, world, port, root, out):
    import os
    import sys
    base = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(base, "p2i-gan-benchmark_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from p2igan_bench import ops
    from p2igan_bench.engine import TrainEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the one GPU of the test box
    dev = torch.device("cuda:0")
    cfg = _zarr_cfg(root)
    _, G, D = _build(dev, 128, 128)
    eng = TrainEngine(G, D, cfg, distributed=True)
    losses = []
    for fr, mk in _zarr_rank_batches(cfg, rank, world, 2):
        r = eng.train_step(*ops.assemble_batch(fr.to(dev).contiguous(), mk.to(dev).contiguous()))
        losses.append({k: float(r[k]) for k in ("loss_g", "loss_d", "pool", "reg")})
    torch.save({"g": eng.gp.flat.cpu(), "d": eng.dp.flat.cpu(), "losses": losses}, os.path.join(out, f"z{rank}.pt"))
    dist.destroy_process_group()


def test_zarr_windows_two_rank_data_parallel_matches_single_process(dev, tmp_path):
    """BASELINE configs[2] on one GPU: a synthetic train.zarr (events/<key>/frames uint8 + index/windows of length 16,
    preprocess.py:195-225, written with zarr_lite) -> P2IDataModule(cfg, rank, 2) -> ShardedSampler -> device_assemble ->
    TrainEngine(distributed=True), 8 samples per rank at 128x128 for two steps, against ONE process stepping on the union of the
    two ranks' batches (global batch 16).  gloo moves the buckets because both ranks share the box's single GPU; on a node the
    same code runs over RCCL (the 8-GPU run itself is the driver's)."""
    import os
    import torch.multiprocessing as mp
    from p2igan_bench import ops
    from p2igan_bench.data.synth_store import write_train_zarr
    from p2igan_bench.engine import TrainEngine
    root = tmp_path / "train.zarr"
    nwin = write_train_zarr(str(root), n_events=5, frames_per_event=30, h=136, w=136, stride=2)     # 136 > 128: random crops
    assert nwin == 40                                                                                # 80/20 split -> 32 training windows
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_zarr_dp_worker, args=(2, port, str(root), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "z0.pt"), torch.load(tmp_path / "z1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["d"], r1["d"])          # ranks stay bit-identical
    cfg = _zarr_cfg(root)
    per_rank = [_zarr_rank_batches(cfg, r, 2, 2) for r in range(2)]
    assert not torch.equal(per_rank[0][0][1], per_rank[1][0][1])                    # the ranks drew different masks
    _, G, D = _build(dev, 128, 128)
    eng = TrainEngine(G, D, cfg)
    for step in range(2):
        fr = torch.cat([per_rank[r][step][0] for r in range(2)]).to(dev)
        mk = torch.cat([per_rank[r][step][1] for r in range(2)]).to(dev)
        out = eng.train_step(*ops.assemble_batch(fr.contiguous(), mk.contiguous()))
        if step == 0:       # per-rank losses are means over 8 samples; their average is the global-batch mean
            for k in ("loss_g", "loss_d", "pool", "reg"):
                avg = 0.5 * (r0["losses"][0][k] + r1["losses"][0][k])
                assert abs(float(out[k]) - avg) <= 2e-5 * abs(avg), (k, float(out[k]), avg)
    # step-1 Adam (beta1 = 0) is a sign update: compare with a budget of a few flipped signs per million weights
    for key, flat in (("g", eng.gp.flat), ("d", eng.dp.flat)):
        diff = (flat.cpu() - r0[key]).abs()
        assert float((diff > 5e-5).float().mean()) < 2e-3, key
        assert float(diff.max()) <= 4.1e-4, key


def test_auto_graph_replay_for_launch_bound_steps(dev, monkeypatch):
    """With P2I_AUTO_GRAPH=1 TrainEngine captures the step by itself once it has repeated AUTO_GRAPH_AFTER times with the same
    shapes, keeps following the eager engine (same tolerance reasoning as test_graph_replay_matches_eager_steps, two steps later),
    and runs a batch of another shape eagerly without disturbing the captured graph or the Adam step counters."""
    from p2igan_bench.engine import TrainEngine
    frames, masked, masks = [t.to(dev) for t in _batch32()]
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("P2I_AUTO_GRAPH", mode)
        cfg, G, D = _build(dev)
        eng = TrainEngine(G, D, cfg)
        for i in range(5):
            r = eng.train_step(frames, masked, masks)
            assert (getattr(eng, "_graph", None) is not None) == (mode == "1" and i >= TrainEngine.AUTO_GRAPH_AFTER), (mode, i)
        runs[mode] = {k: float(r[k]) for k in ("loss_g", "loss_d", "rec")}
        assert eng.opt_g.step_count == 5 and eng.opt_d.step_count == 5
        if mode == "1":
            assert int(eng.opt_g.step_dev) == 5
            r1 = eng.train_step(frames[:1].contiguous(), masked[:1].contiguous(), masks[:1].contiguous())     # tail batch: eager
            assert r1["preds"].shape[0] == 1 and bool(torch.isfinite(r1["loss_g"]))
            assert eng.opt_g.step_count == 6 and int(eng.opt_g.step_dev) == 6
            r2 = eng.train_step(frames, masked, masks)                                                          # replay again
            assert r2["preds"].shape[0] == 2 and int(eng.opt_g.step_dev) == 7
    for k in runs["0"]:
        assert abs(runs["0"][k] - runs["1"][k]) <= 1e-2 * abs(runs["0"][k]), (k, runs)
