#!/usr/bin/env python3
"""Sliding-window inference on the MI355X path; CLI flags, checkpoint resolution and the output
Zarr group follow the reference's scripts/infer.py:20-40,61-80,168-180,247-257."""
from __future__ import annotations

import argparse
import json
import logging
import os
import shutil
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from p2igan_bench.data import zarr_lite  # noqa: E402
from p2igan_bench.data.dataloader import P2IDataModule  # noqa: E402
from p2igan_bench.inference import infer_event  # noqa: E402
from p2igan_bench.models import build_generator  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Inference for P2I-GAN benchmark models")
    p.add_argument("--config", type=Path, default=Path("p2igan_bench/config/p2igan_baseline.json"))
    p.add_argument("--checkpoint", type=Path, default=None)
    p.add_argument("--model-dir", type=Path, default=None)
    p.add_argument("--data-root", type=Path, default=None)
    p.add_argument("--output", type=Path, default=None)
    p.add_argument("--passes", type=int, default=1)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--log-every", type=int, default=50)
    p.add_argument("--stride", type=int, default=16)
    p.add_argument("--overlap", type=int, default=12)
    p.add_argument("--output-scale", type=float, default=255.0)
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--log-level", type=str, default="INFO")
    return p.parse_args(argv)


def resolve_checkpoint(cfg, args) -> Path:
    if args.checkpoint:
        return args.checkpoint
    base = Path(args.model_dir or cfg.get("save_dir", "weights"))
    if base.is_file():
        return base
    if (base / "latest.pt").exists():
        return base / "latest.pt"
    if base.exists():
        cands = sorted(base.glob("*.pt"), key=lambda p: p.stat().st_mtime, reverse=True)
        if cands:
            logging.warning("latest.pt not found, falling back to %s", cands[0])
            return cands[0]
    raise FileNotFoundError(f"Checkpoint not found under {base}")


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO), format="%(asctime)s | %(levelname)s | %(message)s")
    with args.config.open() as f:
        cfg = json.load(f)
    torch.manual_seed(cfg.get("seed", 42))
    np.random.seed(cfg.get("seed", 42))
    if args.data_root is not None:
        cfg.setdefault("data", {}).setdefault("test", {})["data_root"] = str(args.data_root)
    dev = torch.device(args.device or cfg.get("device", "cuda:0"))
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("the MI355X path needs a GPU: there is no CPU fallback")
    ckpt = resolve_checkpoint(cfg, args)
    loader = P2IDataModule(cfg).test_dataloader()
    if loader is None or len(loader.dataset) == 0:
        raise RuntimeError("Test dataloader is not configured or empty.")
    name = cfg.get("model", {}).get("name", "model")
    out = Path(args.output or Path(args.model_dir or cfg.get("save_dir", "weights")) / f"test{name}.zarr")
    if out.exists():
        if not args.overwrite:
            raise FileExistsError(f"Output already exists: {out}")
        shutil.rmtree(out)
    group = zarr_lite.Group(str(out), "w")
    group.attrs.update({"config_path": str(args.config), "checkpoint": str(ckpt), "model_name": name,
                        "data_root": cfg.get("data", {}).get("test", {}).get("data_root"), "passes": int(args.passes),
                        "output_scale": float(args.output_scale)})
    G = build_generator(cfg).to(dev)
    state = torch.load(ckpt, map_location=dev, weights_only=True)
    G.load_state_dict(state["generator"] if isinstance(state, dict) and "generator" in state else state)
    G.eval()
    results = {}
    for p in range(max(1, args.passes)):
        t0 = time.time()
        for i, batch in enumerate(loader):
            _, masked, masks = [t.permute(0, 1, 4, 2, 3).contiguous().to(dev) for t in batch]
            comp = infer_event(G, masked, masks, max(1, args.stride), max(0, args.overlap), args.output_scale).cpu().numpy()
            key = f"event_{i + 1:02d}"
            results[key] = comp if p == 0 else results[key] + (comp - results[key]) / float(p + 1)
            if (i + 1) % max(1, args.log_every) == 0:
                logging.info("Pass %d | %d samples | %.2f samples/sec", p + 1, i + 1, (i + 1) / max(time.time() - t0, 1e-6))
    for key, comp in results.items():
        group.create_dataset(key, comp.astype(np.float32), chunks=comp.shape)
    logging.info("Inference completed. Output saved to %s", out)


if __name__ == "__main__":
    main()
