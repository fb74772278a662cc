#!/usr/bin/env python3
"""Which stream-fork shapes survive hipGraph capture on this ROCm?  (DESIGN.md §4 "capture_end": round 3 saw a segmentation fault at
capture_end when a SideStream was used inside another SideStream.run during capture.)

Each case runs in its OWN child process (a host-side crash must not take the others down) with torch tensors only -- no kernel of
this repo -- so the result is a property of the HIP runtime's capture bookkeeping:

  flat          origin -> A -> origin                      (fork one side stream, join it)
  flat_reuse    origin -> A -> origin, twice               (the same side stream forked twice from the origin)
  nested        origin -> A -> (A -> B -> A) -> origin     (B forked from the NON-origin capturing stream A, joined back into A)
  nested_reuse  nested, then origin -> B -> origin         (B forked from A first, later forked from the origin as well)
  nested_via_origin  B first joins the capture through an event of the ORIGIN stream, then also waits for A's event; joined into A

usage: python tools/graph_nested_fork.py            (prints one line per case: ok / exit code or signal)
"""
import subprocess
import sys

CASES = ("flat", "flat_reuse", "nested", "nested_reuse", "nested_via_origin")


def child(case: str):
    import torch
    dev = torch.device("cuda:0")
    x = torch.ones(1 << 16, device=dev)
    ya, yb = torch.zeros_like(x), torch.zeros_like(x)
    A, Bs = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def fork(src, dst):
        ev = torch.cuda.Event()
        ev.record(src)
        dst.wait_event(ev)

    def body(origin):
        fork(origin, A)
        with torch.cuda.stream(A):
            ya.add_(x)
            if case.startswith("nested"):
                if case == "nested_via_origin":
                    fork(origin, Bs)
                fork(A, Bs)
                with torch.cuda.stream(Bs):
                    yb.add_(x)
                fork(Bs, A)
                ya.add_(yb)
        fork(A, origin)
        if case in ("flat_reuse",):
            fork(origin, A)
            with torch.cuda.stream(A):
                ya.add_(x)
            fork(A, origin)
        if case == "nested_reuse":
            fork(origin, Bs)
            with torch.cuda.stream(Bs):
                yb.add_(x)
            fork(Bs, origin)

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(s)                                    # warm-up, eager
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ya.zero_()
    yb.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(torch.cuda.current_stream())
    g.replay()
    torch.cuda.synchronize()
    print(f"{case}: captured and replayed, ya[0]={float(ya[0])} yb[0]={float(yb[0])}", flush=True)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    for case in CASES:
        r = subprocess.run([sys.executable, __file__, case], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = (r.stderr.strip().splitlines() or [""])[-1][:200]
        print(f"{case:18s} rc={r.returncode:4d}  {tail}  {err if r.returncode else ''}", flush=True)


if __name__ == "__main__":
    main()
