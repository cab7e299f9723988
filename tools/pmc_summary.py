"""Per-kernel mean of a rocprofv3 --pmc counter_collection.csv (FETCH_SIZE / WRITE_SIZE are in KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
name = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-44:]
    acc[k][0] += 1; acc[k][1] += float(r['Counter_Value']); name = r['Counter_Name']
print(name)
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-46s calls %6d  mean %.1f KiB  total %.1f MiB' % (k, n, v / n, v / 1024))
