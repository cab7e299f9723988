"""Per-kernel mean of every counter in a rocprofv3 --pmc counter_collection.csv: python tools/pmc_any.py file.csv [kernel substring]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:] + ' grid=%s' % r.get('Grid_Size', '?')
    if sub in k:
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print('   %-28s n=%4d mean %.4g' % (c, n, v / n))
