"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean of each counter over dispatches (and the dispatch count)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    full = r['Kernel_Name'].split('(')[0]
    if len(sys.argv) > 2 and sys.argv[2] not in full:
        continue
    k = full.replace('void ', '').replace('mf::', '')[:90]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k][r['Counter_Name']] += 1
for k, v in agg.items():
    print(k, {a: round(b / cnt[k][a]) for a, b in v.items()}, 'dispatches', max(cnt[k].values()))
