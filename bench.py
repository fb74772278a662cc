#!/usr/bin/env python3
"""Headline benchmark: P2I-GAN train frames/s on synthetic (B,16,1,128,128) events (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

N>1 either arrives through `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when WORLD_SIZE is unset, bench.py starts the N ranks
itself: the parent spawns N fresh child processes BEFORE touching the GPU (no HIP call, no re-exec), waits for them and
exits non-zero if any of them fails.  A WORLD_SIZE that disagrees with --gpus is an error, never a silent 1-rank run.

One "step" = one full G+D training iteration (train.py:240-326: G fwd, rec loss, D fwd x2 + D bwd +
Adam-D, D fwd + adv loss, G bwd through D, Adam-G) on one batch already resident in HBM.
Per-GPU batch is fixed (weak scaling): configs[1] = B=8 on 1 GPU, configs[2] = 8 per GPU on 8 GPUs.
Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "p2i-gan-benchmark_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T, H, W = 16, 128, 128
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
GFLOP_PER_SAMPLE_STEP = 183.97     # algorithmic conv FLOPs of one full train step


def make_cfg():
    return {"seed": 2024, "model": {"name": "p2igan", "in_channels": 1},
            "data": {"train": {"h": H, "w": W, "sample_length": T}},
            "loss": {"use_gan": 1, "gan_loss": "hinge", "k1_weight": 0.05, "adversarial_weight": 0.01},
            "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}


def cpu_baseline(batch, steps, threads):
    """The CPU oracle (port of the reference step; reference source does not travel) timed on host cores."""
    from oracle import p2i_oracle as orc
    from p2igan_bench.utils import seeded
    torch.set_num_threads(threads)
    orc.IDW_IMPL = "torch"         # time what the reference executes (cdist + topk), not the pinned C checker
    cfg = make_cfg()
    st = orc.TrainState(seeded.seeded_generator_state(H, W, mode="init"), seeded.seeded_discriminator_state(mode="init"),
                        cfg["loss"], cfg["train"]["optimizer"])
    frames, masked, masks = seeded.synthetic_batch(batch, T, H, W, seeded.gauge_mask(H, W, 79))
    st.step(frames, masked, masks)                    # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(frames, masked, masks)
    dt = (time.perf_counter() - t0) / steps
    orc.IDW_IMPL = "c"
    return {"value": batch * T / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{steps} full train steps at B={batch} (T=16,128x128, 79 gauges/frame) after 1 warm-up, {dt:.2f} s/step"}


def launch_ranks(n: int) -> int:
    """Parent side of `--gpus N` without a launcher: N child processes, one per GPU, env contract of
    torch.distributed.run.  The parent never initialises the GPU; rank 0's child prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                       # one rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def physical_cores():
    """(cores this process may use, of which distinct physical cores): affinity mask, cgroup quota, SMT siblings."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    avail = len(cpus)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = set()
    for c in cpus:
        try:
            cores.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip())
        except OSError:
            cores.add(str(c))
    return avail, max(1, min(avail, len(cores)))


def collective_evidence(device=None):
    """What the process group itself saw, for the N > 1 line: the backend's name (nccl = RCCL on ROCm), the number of ranks that
    took part in a SUM all-reduce of ones, and the library version when torch exposes it.  A line whose n_gpus came from the
    environment only cannot show that a collective ran; this does."""
    one = torch.ones(1, device=device) if device is not None else torch.ones(1)
    dist.all_reduce(one)
    ev = {"backend": dist.get_backend(), "ranks_seen": int(one.item()), "world_size": dist.get_world_size()}
    try:
        if device is not None:
            ev["nccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception:
        pass
    return ev


def stub_main(args, world, rank):
    """Launcher self-test (tests/test_bench_launch_cpu.py; P2I_BENCH_STUB=1 only): same rank / barrier / max-over-ranks /
    one-JSON-line plumbing over gloo on CPU with a stand-in step.  Never a measurement: the line says so."""
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.ones(1024) * (rank + 1)
    exch = [0.0]

    def step():
        time.sleep(0.002)
        if world > 1:
            t_ = time.perf_counter()
            dist.all_reduce(buf)
            exch[0] += time.perf_counter() - t_

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    if os.environ.get("P2I_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    rccl = None
    if world > 1:
        rccl = collective_evidence()
        rccl["exchange_ms_per_step"] = exch[0] / max(1, args.steps + args.warmup) * 1e3
    if rank == 0:
        line = {"metric": "train frames/sec (128x128x16)", "value": world * args.batch * T * args.steps / dt, "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "data": "STUB STEP (launcher self-test, not a measurement)",
                "config": {"global_batch": args.batch * world, "parallelism": "dp%d" % world}}
        if rccl is not None:
            line["rccl"] = rccl
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def stack_b32(dev, batch=32, iters=10):
    """north_star target figure: one DO-Conv 3x3 C->C layer of each generator level (C, S) at B=32, forward + dgrad (with
    the residual add) + wgrad, HIP events on the launch stream, against the fp32-matrix peak."""
    from p2igan_bench import ops
    levels, tot_f, tot_s = {}, 0.0, 0.0
    for C, S in ((64, H), (128, H // 2), (256, H // 4), (512, H // 8)):
        spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        x = torch.randn(batch, C, S, S * W // H, device=dev)
        dy = torch.randn_like(x)
        wp_f, wp_d = ops.weight_pack(torch.randn(C, C, 9, device=dev) * 0.05)
        fl = 2.0 * batch * C * C * 9 * x.shape[2] * x.shape[3]
        row = {}
        for kind, fn in (("fwd", lambda: ops.conv_fwd(spec, x, wp_f, act=ops.ACT_RELU)),
                         ("dgrad", lambda: ops.conv_dgrad(spec, dy, wp_d, tuple(x.shape), add=x)),
                         ("wgrad", lambda: ops.conv_wgrad(spec, x, dy))):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / iters
            row[kind] = round(fl / sec / 1e12, 1)
            tot_f += fl
            tot_s += sec
        levels["C%d@%d" % (C, S)] = row
        del x, dy
    ach = tot_f / tot_s / 1e12
    x6 = ops.CONV_ENGINE != "f32"                     # at this batch every launch of the stack is taken by the bf16-split kernels
    peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if x6 else PEAK_FP32_MFMA_TFLOPS
    return {"what": "generator 3x3 DO-Conv stack, one layer per level, fwd+dgrad+wgrad, B=%d; exact-fp32 results; engine %s" % (batch, ops.CONV_ENGINE),
            "peak_note": ("all twelve launches run on the bf16 matrix pipe as six bf16 MFMA products per fp32 product (conv_x6c.hip, wgrad_x6.hip): "
                          "peak = dense bf16 2500 / 6 TF fp32-equivalent; frac_of_f32_mfma_peak prices the same rate against the 157.3 TF the f32 kernels are bound by")
                         if x6 else "f32-MFMA kernels, f32-MFMA peak",
            "bound": "mfma",
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "frac_of_f32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
            "gflop": round(tot_f / 1e9, 1), "ms": round(tot_s * 1e3, 3), "tflops_by_level": levels}


def loader_leg(eng, dev, B, steps, warmup, workers):
    """SURVEY 8d "step time incl. loader": the SAME train step fed by the product's host pipeline instead of a resident batch --
    P2IDataModule on a synthetic train.zarr (configs[2]'s on-disk layout: events/<key>/frames uint8 + index/windows of length
    16, 64 events as SURVEY 8d config 2 prescribes) -> window read + crop + 'sti' mask draw per sample -> uint8 hand-over ->
    ops.assemble_batch on the device (train.device_assemble) -> TrainEngine.train_step.  `workers` = the shipped config's
    train.num_workers unless overridden.  Also times the host pipeline alone (samples/s up to the uint8 batch)."""
    import shutil
    import tempfile
    from p2igan_bench import ops
    from p2igan_bench.data.dataloader import P2IDataModule
    from p2igan_bench.data.synth_store import write_train_zarr
    tmp = tempfile.mkdtemp(prefix="p2i_bench_")
    try:
        root = os.path.join(tmp, "train.zarr")
        nwin = write_train_zarr(root, n_events=64, frames_per_event=30, h=H, w=W, window=T, stride=2)
        cfg = dict(make_cfg(), data={"train": {"data_root": root, "w": W, "h": H, "sample_length": T, "mask": {"type": "sti", "block_sizes": [10]}}})
        cfg["train"] = dict(cfg["train"], batch_size=B, num_workers=workers, device_assemble=True, pin_memory=False, persistent_workers=workers > 0)
        import random
        import numpy as np
        random.seed(cfg["seed"])
        np.random.seed(cfg["seed"])
        loader = P2IDataModule(cfg).train_dataloader()

        from p2igan_bench.data.prefetch import DevicePrefetcher

        def epochs(src):
            while True:
                for b in src:
                    if b[0].shape[0] == B:
                        yield b

        it = epochs(loader)
        next(it)                                          # first batch: worker start-up, page cache
        n_alone = max(8, min(4 * steps, 64))
        t0 = time.perf_counter()
        for _ in range(n_alone):
            next(it)
        host_sps = n_alone * B / (time.perf_counter() - t0)
        del it
        it = epochs(DevicePrefetcher(loader, dev))        # what scripts/train.py iterates: fp32 triples already on the device

        def step():
            eng.train_step(*next(it))

        for _ in range(max(1, warmup)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        del it, loader
        return {"step_ms_with_loader": round(ms, 3), "frames_per_s_with_loader": round(B * T / ms * 1e3, 1),
                "loader_samples_per_s": round(host_sps, 1), "loader_needs_samples_per_s": None,
                "loader": {"store": "synthetic train.zarr, %d windows of %d frames from 64 events (zarr_lite, uncompressed uint8 chunks (20,128,128))" % (nwin, T),
                           "num_workers": workers, "device_assemble": True, "prefetch": "DevicePrefetcher (in line, one batch ahead on a copy stream)",
                           "mask": "sti block 10", "steps": steps}}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def infer_leg(dev, length=40, reps=5):
    """configs[0] / SURVEY 8f3: sliding-window inference (infer.py:188-262: window 16, step 4, last-frame padding, overlap mean) of
    ONE (128,128,L) event through inference.infer_event, all windows of the event in one batched forward.  Training-variant generator
    (DO-Conv folded on the fly, packed weights cached between no-grad forwards) vs P2IGenerator(inference=True) (the reference's
    DOConv2d_eval: pre-folded kernels).  frames/s = event frames delivered per second; windows = generator samples per event."""
    from p2igan_bench.inference import infer_event
    from p2igan_bench.models import build_generator
    from p2igan_bench.models.p2igan import P2IGenerator, fold_generator_state_dict
    from p2igan_bench.utils import seeded
    cfg = make_cfg()
    torch.manual_seed(cfg["seed"])
    G = build_generator(cfg).to(dev).eval()
    Ge = P2IGenerator(cfg, inference=True).to(dev).eval()
    Ge.load_state_dict(fold_generator_state_dict({k: v.detach().clone() for k, v in G.state_dict().items()}), strict=True)
    ev = seeded.synthetic_event(length, H, W, seed=99).float() / 255.0
    frames = ev.reshape(1, length, 1, H, W).to(dev)
    masks = seeded.gauge_mask(H, W, 79).reshape(1, 1, 1, H, W).expand(1, length, 1, H, W).contiguous().to(dev)
    masked = (frames * masks).contiguous()
    out = {"event": "(%d,%d,%d) synthetic event, 79 gauges/frame, window 16 / step 4 -> %d windows in one batch" % (H, W, length, len(range(0, length, 4)))}
    res = {}
    for name, net in (("train_variant", G), ("inference_variant", Ge)):
        for _ in range(2):
            res[name] = infer_event(net, masked, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            infer_event(net, masked, masks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"ms_per_event": round(dt * 1e3, 3), "frames_per_s": round(length / dt, 1), "events_per_s": round(1.0 / dt, 2)}
    a, b = res["train_variant"], res["inference_variant"]
    out["variants_max_rel_diff"] = float((a - b).abs().max() / a.abs().max().clamp(min=1e-30))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)         # (0.6 s of timed steps; 20 / 5 until the end of round 3: the first timed
    ap.add_argument("--warmup", type=int, default=10)        #  steps of a cold process ran 2-3 % slower than the steady state)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (configs[1]: 8)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", choices=("on", "off", "tape"), default="off",
                    help="replay the whole G+D step as one hipGraph (single GPU).  Off by default: at B=8 the step is bound by "
                         "kernel time, not by dispatch gaps (28.6 ms replayed vs 28.8 ms eager, profiles/README.md)")
    ap.add_argument("--no-stack", action="store_true", help="skip the B=32 generator conv-stack figure (roofline_b32_stack)")
    ap.add_argument("--with-loader", action="store_true",
                    help="also time the step fed by P2IDataModule from a synthetic train.zarr through device_assemble (step_ms_with_loader)")
    ap.add_argument("--loader-workers", type=int, default=None, help="DataLoader workers of the --with-loader leg (default: the shipped config's)")
    ap.add_argument("--with-infer", action="store_true", help="also time sliding-window inference of one (128,128,40) event (infer)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                              # no launcher above us: start the N ranks, touch nothing else
            raise SystemExit(launch_ranks(args.gpus))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a mislabelled line")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("P2I_BENCH_STUB") == "1":
        return stub_main(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from p2igan_bench import _hip, ops
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    _hip.load()
    cfg = make_cfg()
    torch.manual_seed(cfg["seed"])                     # reference init (train.py:78-82,121-122), random weights
    G = build_generator(cfg).to(dev)
    D = build_discriminator(cfg).to(dev)
    eng = TrainEngine(G, D, cfg, distributed=world > 1)
    B = args.batch
    mask = seeded.gauge_mask(H, W, 79)
    frames, masked, masks = [t.to(dev) for t in seeded.synthetic_batch(B, T, H, W, mask, seed=2024 + 1000 * rank)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = args.graph in ("on", "tape")
    done = 0
    if use_graph and world == 1 and args.warmup >= 1:
        # the capture's own eager steps are real training steps and count as warm-up; the timed steps are replays of the
        # captured step (same kernels, same arguments, inputs copied into the captured buffers every step)
        done = eng.capture(frames, masked, masks, warmup=min(3, args.warmup), mode="tape" if args.graph == "tape" else "graph")
    for _ in range(max(0, args.warmup - done)):
        eng.train_step(frames, masked, masks)
    sync()
    # one HIP event per step on the launch stream (no host sync inside the timed region): the median step of the run
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        eng.train_step(frames, masked, masks)
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * T * args.steps / dt
    rccl = None
    if world > 1:
        # what RCCL itself saw + the EXPOSED exchange time of a step: HIP events on the launch stream around the D all-reduce and
        # around the wait for the generator's bucketed exchange (TrainEngine.exchange_marks), over a few extra steps outside the
        # timed region
        rccl = collective_evidence(dev)
        eng.exchange_marks = []
        nex = max(2, min(5, args.steps))
        for _ in range(nex):
            eng.train_step(frames, masked, masks)
        torch.cuda.synchronize()
        tot = {"d": 0.0, "g": 0.0}
        for kind, e0, e1 in eng.exchange_marks:
            tot[kind] += e0.elapsed_time(e1)
        eng.exchange_marks = None
        rccl.update(exchange_d_ms_per_step=round(tot["d"] / nex, 3), exchange_g_exposed_ms_per_step=round(tot["g"] / nex, 3),
                    grad_bytes={"d": int(eng.dp.grad.numel() * 4) if eng.dp is not None else 0, "g": int(eng.gp.grad.numel() * 4)},
                    overlap="generator buckets launched per Decoder level inside the backward" if eng.dp_overlap else "flat all-reduce after the backward")

    roofline = None
    extra = {}
    if not args.no_roofline:
        # instrumented pass (HIP events around every conv-engine launch on the launch stream); separate from
        # the timed region above so that `value` is unperturbed.  Every rank steps (the step holds collectives);
        # only rank 0 records.
        if rank == 0:
            ops.PROFILE = ops.KernelProfile()
        graph_saved = getattr(eng, "_graph", None)
        eng._graph = None                              # events cannot be recorded inside a replay: eager launches here
        # The timed region above runs the weight-gradient kernels on a side stream, concurrently with the data-gradient chain; a
        # launch's duration is then the time it SHARED the chip (wgrad 64-ch: ~135 us overlapped vs ~97 us alone).  The per-kernel
        # figures below are meant to judge each kernel against the chip's peak, so this pass keeps every launch on the one stream.
        from p2igan_bench.models import p2igan as _p2igan
        side_saved, _p2igan.SIDE_WGRAD = _p2igan.SIDE_WGRAD, False
        nprof = max(2, min(5, args.steps))
        for _ in range(nprof):
            eng.train_step(frames, masked, masks)
        summ = ops.PROFILE.summary() if rank == 0 else None
        ops.PROFILE = None
        _p2igan.SIDE_WGRAD = side_saved
        # BASELINE.json's second figure, "G-step ms" (G fwd, rec loss, D fwd on fake, adv loss, G bwd incl. the
        # dgrad through D, Adam-G): un-instrumented steps with four phase events each
        # (the step's side-stream overlap is off here: with it D's real-pass forward runs DURING the generator's forward and the
        # phase boundaries on the main stream would charge it to the G step)
        overlap_saved = getattr(eng, "prep_overlap", False)
        eng.prep_overlap = False
        if rank == 0 and eng.use_gan:
            eng.phase_marks = []
        for _ in range(nprof):
            eng.train_step(frames, masked, masks)
        eng.prep_overlap = overlap_saved
        if rank == 0 and eng.use_gan:
            torch.cuda.synchronize()
            m, g_ms, d_ms = eng.phase_marks, 0.0, 0.0
            for i in range(0, len(m) - 3, 4):
                g_ms += m[i].elapsed_time(m[i + 1]) + m[i + 2].elapsed_time(m[i + 3])
                d_ms += m[i + 1].elapsed_time(m[i + 2])
            eng.phase_marks = None
            extra["g_step_ms"] = round(g_ms / nprof, 3)
            extra["d_step_ms"] = round(d_ms / nprof, 3)
            extra["phase_note"] = "g_step_ms / d_step_ms: main-stream phase events of steps run WITHOUT the side-stream overlap of D's real pass / weight preparation (P2I_PREP_OVERLAP=0); ms_per_step is with it"
        eng._graph = graph_saved
    if rank == 0 and not args.no_roofline:
        # dominant kernel = the conv-engine instance with the largest share of the step (single-kernel keys only)
        single = {k: v for k, v in summ.items()
                  if (k.startswith(("patch_gemm_dma_kernel<", "wgrad_dma_kernel<")) and "(" not in k) or k.startswith(("patch_gemm_x6c_kernel<", "patch_gemm_x6p_kernel<", "wgrad_x6_kernel"))}
        dom = max(single, key=lambda k: single[k]["seconds"])
        d = single[dom]
        ach = d["flops"] / d["seconds"] / 1e12
        # the bf16-split kernels execute six bf16 MFMA flops per algorithmic (fp32) flop: their bound is the dense bf16 peak / 6
        x6 = "x6" in dom
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if x6 else PEAK_FP32_MFMA_TFLOPS
        dom_name = dom.split(" (")[0]                  # rocprofv3's kernel name
        # HBM bytes per launch from the committed PMC passes (static: counters cannot be read inside this run).  Corrected as
        # MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 (16-B-per-lane LDS-DMA streams), WRITE_SIZE as read.
        traffic = traffic_src = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            tj = json.load(open(pmc))
            kt = tj.get("kernels", tj)
            traffic = kt.get(dom_name)
            if traffic is None:            # rocprofv3 prints trailing template arguments the plan does not carry (epilogue form, producer waves)
                cand = [v for k_, v in kt.items() if k_.startswith(dom_name.rstrip(">"))]
                traffic = cand[0] if len(cand) == 1 else None
            traffic_src = tj.get("source")
        note = "instrumented pass: every launch on one stream (P2I_SIDE_WGRAD=0), so a launch's duration is the kernel's own"
        if dom.startswith(("wgrad_dma_kernel", "wgrad_x6_kernel")) and ops.WGRAD_SLICES:
            note += "; HIP events bracket the p2i_conv_wgrad_ws call = this kernel + its wgrad_reduce_kernel (rocprofv3 lists them separately)"
        if x6:
            note += ("; exact-fp32 result computed as 6 bf16 MFMA products per fp32 product (3-way bf16 split): achieved = algorithmic fp32 flops / time, "
                     "peak = dense bf16 MFMA peak 2500 / 6; frac_of_f32_mfma_peak prices the same rate against the f32-MFMA peak the f32 kernels are bound by")
        roofline = {"bound": "mfma", "kernel": dom, "note": note, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "frac_of_f32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": round(d["seconds"] / d["launches"] * 1e6, 2),
                    "flops_per_launch": d["flops"] / d["launches"], "launches_per_step": d["launches"] / nprof}
        tot_s = sum(v["seconds"] for v in summ.values())
        tot_f = sum(v["flops"] for v in summ.values())
        extra["conv_engine_all"] = {"tflops": round(tot_f / tot_s / 1e12, 2), "ms_per_step": round(tot_s / nprof * 1e3, 2),
                                    "frac_of_peak": round(tot_f / tot_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        extra["kernels"] = {k: {"ms_per_step": round(v["seconds"] / nprof * 1e3, 3), "tflops": round(v["flops"] / v["seconds"] / 1e12, 2)}
                            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"])}
    if world > 1:
        dist.barrier()

    if rank == 0 and world == 1 and not args.no_stack:
        extra["roofline_b32_stack"] = stack_b32(dev)
    if rank == 0 and world == 1 and args.with_loader:
        shipped = json.load(open(os.path.join(ROOT, "p2i-gan-benchmark_amd", "p2igan_bench", "config", "p2igan_gan_baseline.json")))
        workers = args.loader_workers if args.loader_workers is not None else int(shipped["train"].get("num_workers", 0))
        ll = loader_leg(eng, dev, B, args.steps, args.warmup, workers)
        ll["loader_needs_samples_per_s"] = round(B / (ms_per_step * 1e-3), 1)     # what the resident-batch step rate consumes
        extra.update(ll)
    if rank == 0 and world == 1 and args.with_infer:
        extra["infer"] = infer_leg(dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        avail, phys = physical_cores()                 # all physical cores this process may use
        threads = int(os.environ.get("P2I_CPU_THREADS", phys))
        cpu = cpu_baseline(args.cpu_batch, args.cpu_steps, threads)
        cpu["cores_available"] = avail
        cpu["cores_note"] = "cores = the threads used = every core this process may run on (the lease's share of the host, not the whole socket)"

    if rank == 0:
        line = {"metric": "train frames/sec (128x128x16)", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(ms_median, 3),
                "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "configs[1]: p2igan_gan_baseline train step, hinge GAN, B=%d per GPU, T=16, 128x128, 79 gauge points/frame" % B,
                           "global_batch": B * world, "parallelism": "dp%d" % world,
                           "launch": ("launch tape replay of the whole step (p2i_tape_replay)" if getattr(eng, "_tape", None) is not None else
                                      "hipGraph replay of the whole step") if getattr(eng, "_graph", None) is not None else "eager launches"},
                "step_tflops": round(GFLOP_PER_SAMPLE_STEP * B * world / (ms_per_step * 1e-3) / 1e3, 2),
                "roofline": roofline, "cpu_baseline": cpu}
        if rccl is not None:
            line["rccl"] = rccl
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
