"""Pin the CPU oracle (oracle/p2i_oracle.py) to golden vectors captured from the genuine
reference by tests/golden/make_golden.py.  CPU only."""
import numpy as np
import torch

from conftest import rel_err
from oracle import p2i_oracle as O
from p2igan_bench.utils import seeded

TOL = 1e-4   # north_star: 1e-4 rel fp32
# Adam with beta1=0 moves every weight by ~lr*sign(g) on step 1: a gradient whose sign flips under
# fp32 summation-order noise moves a weight by 2e-4, so parameter checksums get an absolute budget.
PSUM_TOL = 2e-3


def _batch32():
    h = w = 32
    m0 = seeded.gauge_mask(h, w, 20)
    m1 = seeded.block_mask(h, w, 4)
    f0, k0, mk0 = seeded.synthetic_batch(1, 16, h, w, m0, seed=2024)
    f1, k1, mk1 = seeded.synthetic_batch(1, 16, h, w, m1, seed=3024)
    return torch.cat([f0, f1]), torch.cat([k0, k1]), torch.cat([mk0, mk1])


def test_idw_matches_reference(golden):
    """N >= 256 (torch.topk -> std::partial_sort, emulated exactly by oracle/idw_knn.c): bit-level
    agreement with the reference.  N < 256 (nth_element path): the order of EXACT rank-4/5 distance
    ties is unpinned; everywhere else the results agree."""
    g = golden("idw.npz")
    for name in ("gauge", "few", "lattice"):
        mask = torch.from_numpy(g[name + "_mask"])
        mk = mask.reshape(1, 32, 32).expand(16, 32, 32)
        tz, ty, tx, pts = O.mask_points(mk)
        out = O.idw_3d_knn(pts, torch.from_numpy(g[name + "_vals"]), (16, 32, 32))
        if pts.shape[0] >= 256:
            assert rel_err(out.numpy(), g[name + "_out"]) < 1e-6, name
        else:
            d = torch.cdist(O.grid_points(16, 32, 32), pts).sort(dim=1)[0]
            no_tie = (d[:, 3] != d[:, 4]).reshape(16, 32, 32).numpy()
            assert no_tie.mean() > 0.5
            err = np.abs(out.numpy() - g[name + "_out"])
            assert err[no_tie].max() < 1e-6 * np.abs(g[name + "_out"]).max(), name


def test_c_selection_equals_torch_topk_on_this_host():
    """oracle/idw_knn.c reproduces torch.cdist + torch.topk (set of 4 selected points per voxel) on the
    host the goldens were captured on (Intel AVX-512: MKL sgemm = k-ordered fmaf chain)."""
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "idw.npz"))
    for name in ("gauge", "lattice"):
        mask = torch.from_numpy(g[name + "_mask"])
        tz, ty, tx, pts = O.mask_points(mask.reshape(1, 32, 32).expand(16, 32, 32))
        vals = torch.from_numpy(g[name + "_vals"])
        _, sel_t = O.idw_3d_knn_torch(pts, vals, (16, 32, 32), return_sel=True)
        sel_c, _ = O.idw_select_c(pts, (16, 32, 32))
        same = (sel_c.sort(1)[0] == sel_t.sort(1)[0]).all(1).float().mean().item()
        if same < 1.0:
            import pytest
            pytest.skip(f"host CPU's MKL path differs from the golden host ({same:.5f} of voxels agree)")


def test_train_steps_match_reference(golden):
    g = golden("e2e_32.npz")
    frames, masked, masks = _batch32()
    st = O.TrainState(seeded.seeded_generator_state(32, 32), seeded.seeded_discriminator_state(),
                      {"k1_weight": 0.05, "adversarial_weight": 0.01, "gan_loss": "hinge"},
                      {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99})
    r = st.step(frames, masked, masks, keep_grads=True)
    assert rel_err(r["preds"].numpy(), g["preds"]) < TOL
    assert rel_err(r["logits_fake"].numpy(), g["logits_fake"]) < TOL
    assert rel_err(r["logits_real"].numpy(), g["logits_real"]) < TOL
    for k in ("loss_g", "loss_d", "adv", "pool", "reg"):
        assert abs(r[k] - float(g[k])) <= TOL * abs(float(g[k])), k
    for k in g.files:
        if k.startswith("ggradnorm/"):
            n = k.split("/", 1)[1]
            assert abs(float(r["ggrads"][n].norm()) - float(g[k])) <= 1e-3 * float(g[k]) + 1e-7, k
        if k.startswith("dgradnorm/"):
            n = k.split("/", 1)[1]
            assert abs(float(r["dgrads"][n].norm()) - float(g[k])) <= 1e-3 * float(g[k]) + 1e-7, k
        if k.startswith("ggrad/"):
            assert rel_err(r["ggrads"][k.split("/", 1)[1]].numpy(), g[k]) < 1e-3, k
        if k.startswith("dgrad/"):
            assert rel_err(r["dgrads"][k.split("/", 1)[1]].numpy(), g[k]) < 1e-3, k
    assert r["dgrads"]["alpha3d"] is None      # alpha3d never receives a gradient (p2igan.py:145,170)
    for k in g.files:
        if k.startswith("g1sum/"):
            n = k.split("/", 1)[1]
            assert abs(float(st.gp[n].double().sum()) - float(g[k])) <= PSUM_TOL * max(1.0, abs(float(g[k]))), k
        if k.startswith("d1sum/"):
            n = k.split("/", 1)[1]
            assert abs(float(st.dp[n].double().sum()) - float(g[k])) <= PSUM_TOL * max(1.0, abs(float(g[k]))), k
    assert rel_err(st.dp["d3d.0.weight_u"].numpy(), g["d1/d3d.0.weight_u"]) < 1e-5
    r = st.step(frames, masked, masks)
    assert abs(r["loss_g"] - float(g["loss_g_step1"])) < 1e-3 * abs(float(g["loss_g_step1"]))
    r = st.step(frames, masked, masks)
    assert abs(r["loss_g"] - float(g["loss_g_step2"])) < 1e-3 * abs(float(g["loss_g_step2"]))
    assert abs(r["loss_d"] - float(g["loss_d_step2"])) < 1e-3 * abs(float(g["loss_d_step2"]))
    assert rel_err(st.gp["Convsin.0.main.0.W"].numpy(), g["g3/Convsin.0.main.0.W"]) < 1e-3
    assert rel_err(st.dp["d2d.2.bias"].numpy(), g["d3/d2d.2.bias"]) < 1e-3


def _oracle_full_step(golden, name, h):
    import fullsize
    g = golden(name)
    frames, masked, masks = fullsize.batch_for(g, h, h)
    st = O.TrainState(seeded.seeded_generator_state(h, h), seeded.seeded_discriminator_state(),
                      {"k1_weight": 0.05, "adversarial_weight": 0.01, "gan_loss": "hinge"}, {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99})
    taps = {}
    r = st.step(frames, masked, masks, keep_grads=True, taps=taps)
    fullsize.check(g, r, r["ggrads"], r["dgrads"], st.gp, st.dp, {k: v.detach() for k, v in taps.items()})
    assert r["dgrads"]["alpha3d"] is None


def test_full_size_train_step_128_matches_reference(golden):
    """configs[1] geometry (B=2, 128x128): one full G+D step of the oracle vs the genuine reference's (e2e_128.npz)."""
    _oracle_full_step(golden, "e2e_128.npz", 128)


def test_full_size_train_step_256_matches_reference(golden):
    """configs[3] geometry (B=1, 256x256, 316 gauges): e2e_256.npz."""
    _oracle_full_step(golden, "e2e_256.npz", 256)


def test_infer_event_matches_reference(golden):
    g = golden("infer_32.npz")
    h = w = 32
    gp = seeded.seeded_generator_state(h, w)
    m = seeded.gauge_mask(h, w, 24, seed=7)
    L = 40
    ev = seeded.synthetic_event(L, h, w, seed=99).float() / 255.0
    frames = ev.reshape(1, L, 1, h, w)
    masks = m.reshape(1, 1, 1, h, w).expand(1, L, 1, h, w).contiguous()
    comp = O.infer_event(gp, frames * masks, masks)
    assert rel_err(comp.numpy(), g["comp"]) < TOL
    with torch.no_grad():
        z = O.generator_forward(gp, frames[:, :16] * 0, masks[:, :16] * 0)
        logits = O.discriminator_forward(seeded.seeded_discriminator_state(), frames[:, :16], training=False)
    assert rel_err(z.numpy(), g["empty"]) < TOL
    assert rel_err(logits.numpy(), g["logits_eval"]) < TOL


def test_generator_128_matches_reference(golden):
    g = golden("g_128.npz")
    gp = seeded.seeded_generator_state(128, 128)
    frames, masked, masks = seeded.synthetic_batch(1, 16, 128, 128, seeded.gauge_mask(128, 128, 79))
    taps = {}
    with torch.no_grad():
        preds = O.generator_forward(gp, masked, masks, taps)
    assert rel_err(taps["idw"].numpy()[0, :, ::3, ::3], g["idw_s3"]) < 1e-6
    assert rel_err(preds.numpy()[0, :, 0, ::3, ::3], g["preds_s3"]) < TOL
    assert abs(float(preds.double().sum()) - float(g["preds_sum"])) < 1e-4 * float(g["preds_abs_sum"])


def test_metrics_oracle_matches_reference_golden(golden):
    """oracle/metrics_oracle.py vs the reference's own metric classes (metrics_32.npz: two update() calls + compute())."""
    from oracle import metrics_oracle as mo
    from p2igan_bench.utils import seeded
    g = golden("metrics_32.npz")
    thr, scales = (0.5, 2.0, 4.0, 8.0), (1, 2, 4, 8)
    table = torch.zeros(4, 4, dtype=torch.int64)
    fss_sum = torch.zeros(4, 4)
    abs_sum = sq_sum = 0.0
    for i, seed in enumerate((11, 12)):
        p, t = seeded.metric_fields(seed)
        table += mo.contingency(p, t, thr)
        fss_sum += mo.fss(p, t, thr, scales)
        d = mo.transform(p) - mo.transform(t)
        abs_sum += float(d.abs().sum()); sq_sum += float((d ** 2).sum())
        assert np.allclose(fss_sum.numpy(), g[f"fss_after_{i}"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(table.numpy(), g["table"].astype(np.int64))             # integer work: exact
    assert abs(abs_sum - float(g["abs_sum"])) <= 1e-6 * float(g["abs_sum"])
    assert abs(sq_sum - float(g["squared_sum"])) <= 1e-6 * float(g["squared_sum"])
    vals = dict(zip([str(k) for k in g["keys"]], g["values"]))
    cat = mo.categorical_scores(table, thr)
    for k, v in cat.items():
        assert abs(v - vals[k]) <= 1e-6, k
    for ti, th in enumerate(thr):
        for si, sc in enumerate(scales):
            assert abs(float(fss_sum[ti, si] / 2) - vals[f"fss_thr{th:.2f}_s{sc}"]) <= 1e-6
