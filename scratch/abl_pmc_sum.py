import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if ("fast_kernel" in r["Kernel_Name"] and r["Kernel_Name"].split(">(")[0].endswith(", 0"))]
byc={}
for r in rows: byc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
names=("warm","full","-build","-eval","-maskwrite","-obswrite","-phase2","-phase1","-stats","+fuseddraw")
for c,v in byc.items():
    print(c, len(v))
    base=None
    for i,n in enumerate(names):
        seg=v[i*12+2:(i+1)*12]
        if not seg: continue
        m=sum(seg)/len(seg)
        if n=="full": base=m
        print("  %-12s %12.0f %s"%(n,m, "" if base is None else "(%+.0f)"%(m-base)))
