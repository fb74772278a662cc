import csv, collections, glob, sys
d=sys.argv[1]; nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 7
f=(glob.glob(d+"/*kernel_trace.csv")+glob.glob(d+"/*/*kernel_trace.csv"))[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void p2i::","").replace("p2i::","")
    if "wgrad" in n or "patch_gemm" in n:
        key=(n[:46], r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"])
    else: key=(n[:46],"","","")
    agg[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=sum(sum(v) for v in agg.values())
print("total kernel ms/step", tot/1e3/nsteps)
fam=collections.defaultdict(float)
for k,v in agg.items(): fam[k[0].split("<")[0]]+=sum(v)/1e3/nsteps
print({k:round(v,2) for k,v in sorted(fam.items(), key=lambda kv:-kv[1])[:14]})
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv)>3 else 30]:
    print("%-50s %8s %3s %3s n/step %5.1f avg_us %8.1f ms/step %6.2f"%(k[0],k[1],k[2],k[3],len(v)/nsteps,sum(v)/len(v),sum(v)/1e3/nsteps))
