"""How much of a train step is the GPU idle (no kernel running on any stream)?  From a rocprofv3 --kernel-trace csv of
`python3 bench.py --steps K --warmup W --no-cpu-baseline --no-roofline --no-stack`: the union of the kernel intervals over the last
K steps' span against the span itself, and the largest gaps with the kernels around them.
usage: python tools/gpu_idle_from_trace.py <kernel_trace.csv> [steps=5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
# a step ends with the generator's Adam (every second adam launch); take the last K steps
ends = adam[1::2]
lo, hi = ev[ends[-K - 1]][1], ev[ends[-1]][1]
sel = [e for e in ev if e[0] >= lo and e[1] <= hi]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
span = hi - lo
print(f"{K} steps: span {span / K / 1e6:.3f} ms/step, GPU busy (union of kernels) {busy / K / 1e6:.3f} ms/step = {busy / span:.1%}; kernel-time sum {sum(e - s for s, e, _ in sel) / K / 1e6:.3f} ms/step; {len(sel) / K:.0f} kernels/step")
gaps.sort(reverse=True)
tot_gap = sum(g for g, _, _ in gaps)
print(f"idle {tot_gap / K / 1e3:.0f} us/step in {len(gaps) / K:.0f} gaps/step; gaps > 5 us: {sum(g for g, _, _ in gaps if g > 5000) / K / 1e3:.0f} us/step")
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:7.1f} us  after {a[:60]:60s} before {b[:60]}")
