"""Host-side data logic on CPU: mask semantics (sti_dataset.py:18-122), zarr-lite round trip, data module."""
import numpy as np
import torch


def test_create_mask_semantics(tmp_path):
    from p2igan_bench.data.sti_dataset import create_mask
    v = torch.zeros(16, 32, 40, 1)
    np.random.seed(0)
    m = create_mask(v, "sti", block_sizes=[10])
    assert m.shape == v.shape and torch.equal(m[0], m[15])
    assert int(m[0].sum()) == 4 * 4                       # ceil(32/10) * ceil(40/10) cells, one pixel each
    m = create_mask(v, "nowcasting", keep=4)
    assert float(m[:4].min()) == 1 and float(m[4:].max()) == 0
    np.random.seed(1)
    m = create_mask(v, "fi", interval=[3])
    assert [int(m[t].max()) for t in range(16)] == [1 if t % 4 == 0 else 0 for t in range(16)]
    np.random.seed(2)
    m = create_mask(v, "stin", block_sizes=[8], keep=2)
    assert float(m[:2].min()) == 1 and int(m[5].sum()) == 4 * 5
    f = tmp_path / "mask.txt"
    mat = (np.random.rand(32, 40) > 0.9).astype(np.float32)
    np.savetxt(f, mat)
    m = create_mask(v, "stis", mask_file=str(f))
    assert torch.equal(m[7, :, :, 0], torch.from_numpy(mat))


def test_zarr_lite_roundtrip_and_train_store(tmp_path):
    from p2igan_bench.data import zarr_lite
    from p2igan_bench.data.dataloader import P2IDataModule
    root = tmp_path / "train.zarr"
    g = zarr_lite.Group(str(root), "w")
    ev = g.require_group("events")
    rng = np.random.default_rng(0)
    for k in ("201801010000", "201801020000"):
        e = ev.require_group(k)
        e.create_dataset("frames", rng.integers(0, 255, (24, 40, 40), dtype=np.uint8), chunks=(20, 16, 16), compress=(k[-5] == "2"))
    idx = np.array([[0, 0, 16], [0, 4, 16], [1, 8, 16], [1, 2, 16], [0, 8, 16]], dtype=np.int32)
    g.require_group("index").create_dataset("windows", idx)
    back = zarr_lite.Group(str(root))["events"]["201801020000"]["frames"]
    assert back.shape == (24, 40, 40) and back[3:20, 5:33, 7].shape == (17, 28)
    cfg = {"seed": 1, "data": {"train": {"data_root": str(root), "w": 32, "h": 32, "sample_length": 16, "mask": {"type": "sti", "block_sizes": [8]}}},
           "train": {"batch_size": 2, "num_workers": 0}}
    dm = P2IDataModule(cfg)
    video, masked, mask = next(iter(dm.train_dataloader()))
    assert video.shape == (2, 16, 32, 32, 1) and torch.equal(masked, video * mask)
    assert len(dm.train_dataset) + len(dm.valid_dataset) == 5


def test_synthetic_module_and_test_split_drops_sample_length():
    from p2igan_bench.data.dataloader import P2IDataModule
    cfg = {"data": {"train": {"data_root": "synthetic://4", "w": 32, "h": 32, "sample_length": 16, "mask": {"type": "sti", "block_sizes": [8]}},
                    "test": {"data_root": "synthetic://2", "synthetic_length": 40, "sample_length": None}},
           "train": {"batch_size": 2, "num_workers": 0}}
    dm = P2IDataModule(cfg)
    assert next(iter(dm.train_dataloader()))[0].shape == (2, 16, 32, 32, 1)
    assert next(iter(dm.test_dataloader()))[0].shape == (1, 40, 32, 32, 1)


def _sti_matrix_reference_loop(H, W, block):
    """Literal restatement of the per-cell scalar draws of sti_dataset.py:44-58."""
    m = np.zeros((H, W), dtype=np.float32)
    for h0 in range(0, H, block):
        for w0 in range(0, W, block):
            rh = np.random.randint(h0, min(h0 + block, H))
            rw = np.random.randint(w0, min(w0 + block, W))
            m[rh, rw] = 1.0
    return m


def test_vectorised_sti_mask_consumes_numpy_rng_like_the_reference():
    from p2igan_bench.data.sti_dataset import _sti_matrix
    for (H, W, b) in [(128, 128, 4), (128, 128, 10), (32, 48, 5), (256, 256, 7), (20, 22, 3)]:
        np.random.seed(7)
        a = _sti_matrix_reference_loop(H, W, b)
        sa = np.random.randint(0, 1 << 30)
        np.random.seed(7)
        c = _sti_matrix(H, W, b)
        sc = np.random.randint(0, 1 << 30)
        assert np.array_equal(a, c) and sa == sc, (H, W, b)


def test_device_assemble_mode_ships_uint8_with_identical_masks(tmp_path):
    """train.device_assemble: the loader returns (uint8 frames, uint8 mask) drawn with the same RNG calls; the float
    triple is what ops.assemble_batch rebuilds on the GPU (tests/test_ops_gpu.py)."""
    from p2igan_bench.data import zarr_lite
    from p2igan_bench.data.dataloader import P2IDataModule
    root = tmp_path / "train.zarr"
    g = zarr_lite.Group(str(root), "w")
    e = g.require_group("events").require_group("201801010000")
    e.create_dataset("frames", np.random.default_rng(0).integers(0, 255, (24, 32, 32), dtype=np.uint8), chunks=(20, 16, 16))
    g.require_group("index").create_dataset("windows", np.array([[0, 0, 16], [0, 4, 16], [0, 8, 16]], dtype=np.int32))
    base = {"seed": 1, "data": {"train": {"data_root": str(root), "w": 32, "h": 32, "sample_length": 16, "mask": {"type": "sti", "block_sizes": [8]}}},
            "train": {"batch_size": 1, "num_workers": 0}}
    ds_f = P2IDataModule(base).train_dataset
    raw = dict(base, train=dict(base["train"], device_assemble=True))
    ds_u = P2IDataModule(raw).train_dataset
    np.random.seed(3)
    video, masked, mask = ds_f[0]
    np.random.seed(3)
    fr_u8, mk_u8 = ds_u[0]
    assert fr_u8.dtype == torch.uint8 and mk_u8.dtype == torch.uint8 and fr_u8.shape == (16, 32, 32)
    assert torch.equal(fr_u8.float() / 255.0, video[..., 0]) and torch.equal(mk_u8.float(), mask[..., 0])
    assert torch.equal((fr_u8.float() / 255.0) * mk_u8.float(), masked[..., 0])


def test_worker_seeding_varies_per_epoch_and_rank():
    """with num_workers > 0 and persistent_workers off the workers restart every epoch; their numpy / python
    generators must NOT restart from the same seed (torch's default worker seeding, which the reference relies on, differs per
    epoch).  Two epochs over the same samples must draw different 'sti' masks; two ranks must differ too."""
    from p2igan_bench.data.dataloader import P2IDataModule
    cfg = {"seed": 5, "data": {"train": {"data_root": "synthetic://4", "w": 32, "h": 32, "sample_length": 16, "mask": {"type": "sti", "block_sizes": [8]}}},
           "train": {"batch_size": 2, "num_workers": 2, "persistent_workers": False, "pin_memory": False}}

    def epoch_masks(rank, world):
        torch.manual_seed(123)                         # same base-seed stream for both ranks: only the rank term separates them
        dl = P2IDataModule(cfg, rank, world)._loader(P2IDataModule(cfg, rank, world).train_dataset, False, 2)
        return [torch.cat([b[2] for b in dl]) for _ in range(2)]

    e0, e1 = epoch_masks(0, 1)
    assert e0.shape == e1.shape == (4, 16, 32, 32, 1)
    assert not torch.equal(e0, e1)                     # second epoch: new masks
    r1 = epoch_masks(1, 1)[0]
    assert not torch.equal(e0, r1)                     # other rank, same torch seed: different masks


def test_eval_sharding_sees_every_sample_once():
    from p2igan_bench.parallel import ShardedSampler
    for n, world in [(10, 4), (7, 2), (3, 4), (16, 8)]:
        seen = []
        for r in range(world):
            s = ShardedSampler(n, r, world, shuffle=False, even=False)
            idx = list(s)
            assert len(idx) == len(s)
            seen += idx
        assert sorted(seen) == list(range(n))
        tr = [list(ShardedSampler(n, r, world, shuffle=True, seed=3)) for r in range(world)]
        assert len({len(t) for t in tr}) == 1 and len(set(sum(tr, []))) == sum(map(len, tr))     # training shards: equal, disjoint


def test_zarr_chunk_cache_is_one_bounded_lru_per_process(tmp_path, monkeypatch):
    """zarr_lite keeps decoded chunks in ONE least-recently-used cache per process, shared by every Array (round 3 kept up to 64 MB
    per array and a dataset keeps every opened event's Array: host memory grew with events x workers)."""
    import numpy as np
    from p2igan_bench.data import zarr_lite
    g = zarr_lite.Group(str(tmp_path / "s.zarr"), mode="w")
    arrs = [g.create_dataset(f"a{i}", np.full((4, 64, 64), i, dtype=np.float32), chunks=(1, 64, 64)) for i in range(6)]     # 16 KB chunks
    zarr_lite.Array.cache_clear()
    monkeypatch.setattr(zarr_lite.Array, "CACHE_BYTES", 5 * 16384)
    for a in arrs:
        for t in range(4):
            assert float(a[t, 0, 0]) == float(arrs.index(a))
    assert zarr_lite.Array._lru_bytes <= 5 * 16384 and len(zarr_lite.Array._lru) == 5
    keys = list(zarr_lite.Array._lru)
    assert all(k[0].endswith("a5") or k[0].endswith("a4") for k in keys)            # the most recently read chunks survive
    _ = arrs[4][3, 0, 0]                                                            # a hit moves the chunk to the young end
    assert list(zarr_lite.Array._lru)[-1][0].endswith("a4")
    zarr_lite.Array.cache_clear()
