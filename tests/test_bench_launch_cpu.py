"""bench.py --gpus N without a launcher must start N ranks itself (VERDICT r1 #1 / ADVICE): exercised here over gloo with
the stand-in step of P2I_BENCH_STUB=1 (CPU container: no GPU call is reachable on this path)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(P2I_BENCH_STUB="1", **env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out               # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 16
    assert line["steps"] == 5 and line["ms_per_step"] >= 2.0
    # the N > 1 line carries what the process group itself saw (bench.py::collective_evidence), not only WORLD_SIZE
    assert line["rccl"]["backend"] == "gloo" and line["rccl"]["ranks_seen"] == 2 and line["rccl"]["world_size"] == 2
    assert line["rccl"]["exchange_ms_per_step"] >= 0.0


def test_gpus1_stays_single_process():
    r = _run(["--gpus", "1", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and "rccl" not in line


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], P2I_BENCH_STUB_FAIL_RANK="1")
    assert r.returncode != 0


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "2"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_PORT="29999")
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_under_torchrun_contract():
    """The driver's launch for N>1: torch.distributed.run sets the env; bench.py must not spawn again."""
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(P2I_BENCH_STUB="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + os.getpid() % 300), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2
