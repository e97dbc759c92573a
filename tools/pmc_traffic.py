"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md HBM section).

    python tools/pmc_traffic.py <dir_fetch> <dir_write> [kernel substring]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read, so the
read side is doubled (the guide's correction).  Prints mean bytes per launch.
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                acc[name].append(float(r['Counter_Value']))
    return acc


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
pat = sys.argv[3] if len(sys.argv) > 3 else ''
out = {}
for k in sorted(set(fetch) | set(write)):
    if pat and pat not in k:
        continue
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
    out[k] = {'launches': len(fetch.get(k, [])), 'fetch_bytes_raw': f * 1024, 'read_bytes_corrected': 2 * f * 1024,
              'write_bytes': w * 1024, 'hbm_bytes_per_launch': (2 * f + w) * 1024}
print(json.dumps(out, indent=1))
