"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if pat and pat not in name:
            continue
        short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f'   {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}')
