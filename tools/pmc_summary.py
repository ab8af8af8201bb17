"""Summarise the counter_collection CSVs of tools/pmc_all.sh for one kernel: per-launch averages."""
import collections, csv, glob, sys
root, needle = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for f in sorted(glob.glob(root + '/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if needle in r['Kernel_Name']:
            a = acc.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
print(f"kernel filter: {needle}")
for k, (n, v) in acc.items():
    print(f"{k:28s} launches={n:4d} per_launch={v / n:18.0f}")
