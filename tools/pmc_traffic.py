"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command).
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> <out.json> <source tag>
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  Correction of MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half the bytes of 16-B-per-lane streams (global_load_dwordx4 and buffer_load ... lds alike) -> x2 for the conv-engine kernels,
whose patch / weight / x / dy images all arrive by 16-B LDS-DMA; WRITE_SIZE is taken as read.  Both raw columns are kept."""
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
    fo.write("kernel,dispatches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg,HBM_bytes_per_launch_corrected(2*FETCH+WRITE)\n")
    for k, n, fk, wk in rows:
        fo.write('"%s",%d,%.0f,%.0f,%d\n' % (k, n, fk, wk, int((2 * fk + wk) * 1024)))
# template arguments added after round 1 (LJU, SX) are dropped from the key so that bench.py's kernel names match
def key(k):
    if k.startswith("wgrad_dma_kernel<"):
        a = k[k.index("<") + 1:k.rindex(">")].split(", ")
        return "wgrad_dma_kernel<%s>" % ", ".join(a[:4])
    return k
kern = collections.defaultdict(lambda: [0.0, 0])
for k, n, fk, wk in rows:
    if k.startswith(("patch_gemm", "wgrad", "c1_")):
        e = kern[key(k)]
        e[0] += (2 * fk + wk) * 1024 * n
        e[1] += n
json.dump({"source": sys.argv[5] if len(sys.argv) > 5 else sys.argv[3],
           "correction": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md (16-B-per-lane streams read at 1/2)",
           "kernels": {k: int(v[0] / v[1]) for k, v in kern.items()}}, open(sys.argv[4], "w"), indent=1)
print("kernels:", len(rows))
