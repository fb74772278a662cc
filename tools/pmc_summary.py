"""rocprofv3 --pmc counter_collection.csv -> one row per kernel: dispatches and the per-dispatch average of every counter.
usage: python tools/pmc_summary.py <counter_collection.csv> [more csv ...] > out.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("p2i::", "")
        k = k[:k.rfind("(")] if "(" in k else k
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get(names[0], [0])))):
    nd = max(len(x) for x in v.values())
    print('"%s",%d,' % (k, nd) + ",".join("%.0f" % (sum(v[c]) / len(v[c])) if c in v else "" for c in names))
