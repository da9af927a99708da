import csv, collections, sys, glob
f=sorted(glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(set)
for r in rows:
    k=r['Kernel_Name'][:34]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,c in agg.items():
    if 'gemm' not in k: continue
    n=len(cnt[k])
    print(k, "dispatches", n, {a: f"{b/n:.4g}" for a,b in sorted(c.items())})
