"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command).
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  FETCH_SIZE is kept RAW: MI355X_MICROARCH.md calibrates it (x2) only for
16-B-per-lane streams; the conv engine's patch gathers are 4 B per lane."""
import collections
import csv
import json
import sys


def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("p2i::", "")
        k = k[:k.rfind("(")] if "(" in k else k
        agg[k].append(float(r["Counter_Value"]))
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
    fk = sum(f[k]) / len(f[k]) if k in f else 0.0
    wk = sum(w[k]) / len(w[k]) if k in w else 0.0
    rows.append((k, len(f.get(k, w.get(k, []))), fk, wk))
with open(sys.argv[3], "w") as fo:
    fo.write("kernel,dispatches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg\n")
    for k, n, fk, wk in rows:
        fo.write('"%s",%d,%.0f,%.0f\n' % (k, n, fk, wk))
json.dump({k: int((fk + wk) * 1024) for k, n, fk, wk in rows if k.startswith(("patch_gemm", "wgrad", "c1_"))}, open(sys.argv[4], "w"), indent=1)
print("kernels:", len(rows))
