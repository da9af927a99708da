"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_parse.py <dir> [substring]"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))[-1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r['Dispatch_Id'])
for k, c in agg.items():
    if sub and sub not in k:
        continue
    n = len(cnt[k])
    print(k, "dispatches", n, {a: f"{b/n:.4g}" for a, b in sorted(c.items())})
