"""N>1 path on CPU: world_size-2 gloo processes run the flat-bucket gradient exchange and the sharded
sampler; the 2-rank result must equal the 1-rank result at the same global batch."""
import os
import sys

import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(12, 16), nn.Tanh(), nn.Linear(16, 3))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, out):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from p2igan_bench import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    m = _model()
    if rank != 0:                                     # ranks start different; broadcast must align them
        for p in m.parameters():
            p.data.add_(1.0)
    fp = parallel.FlatParams(m)
    parallel.broadcast_module_state(m, fp)
    x, y = _data()
    samp = parallel.ShardedSampler(8, rank, world, shuffle=True, seed=3)
    idx = list(samp)
    for step in range(3):
        fp.zero_grad()
        loss = ((m(x[idx]) - y[idx]) ** 2).mean()     # per-rank batch mean
        loss.backward()
        parallel.allreduce_mean_(fp.grad, world)
        fp.flat.add_(fp.grad, alpha=-0.1)
    torch.save({"flat": fp.flat.clone(), "idx": idx}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                       # ranks stay bit-identical
    assert sorted(r0["idx"] + r1["idx"]) == list(range(8))           # shards partition the epoch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd"))
    from p2igan_bench import parallel
    m = _model()
    fp = parallel.FlatParams(m)
    x, y = _data()
    for step in range(3):
        fp.zero_grad()
        ((m(x) - y) ** 2).mean().backward()                          # 1 rank, global batch 8
        fp.flat.add_(fp.grad, alpha=-0.1)
    assert torch.allclose(fp.flat, r0["flat"], rtol=1e-5, atol=1e-6)


def test_flat_params_views_and_grad_accumulation():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd"))
    from p2igan_bench import parallel
    m = _model()
    ref = [p.detach().clone() for p in m.parameters()]
    fp = parallel.FlatParams(m)
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p.data, r)
    x, y = _data()
    fp.zero_grad()
    ((m(x) - y) ** 2).mean().backward()
    assert float(fp.grad.abs().sum()) > 0                            # autograd accumulated INTO the flat buffer
    m.load_state_dict({k: v + 1 for k, v in m.state_dict().items()})  # in-place load keeps the views
    assert torch.equal(fp.flat[:12 * 16].view(16, 12), m[0].weight.data)


def _bucket_worker(rank, world, port, out):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "p2i-gan-benchmark_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from p2igan_bench import parallel
    parallel.init_distributed("gloo")
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(10_000, generator=g)
    flat = grad.clone()
    parallel.allreduce_mean_(flat, world)
    bk = parallel.BucketedAllReduce(grad, world)
    bk.launch(4000, 7000)                   # buckets become ready out of order, with gaps between them
    bk.launch(100, 900)
    bk.finish()                             # launches [0,100), [900,4000), [7000,10000), waits, scales
    torch.save({"flat": flat, "bucketed": grad}, os.path.join(out, f"b{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_bucketed_allreduce_is_bit_identical_to_flat(tmp_path):
    """The overlapped per-level exchange of the generator's gradients (parallel.BucketedAllReduce, engine.py) must give exactly
    what the single flat all-reduce gives: with two ranks every element is a sum of two addends, so there is no order to differ in."""
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    assert torch.equal(r0["flat"], r0["bucketed"]) and torch.equal(r1["flat"], r1["bucketed"])
    assert torch.equal(r0["bucketed"], r1["bucketed"])
