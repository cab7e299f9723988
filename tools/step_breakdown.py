"""Per-position kernel durations inside a decoder step from a rocprofv3 kernel_trace.csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dec = [r for r in rows if 'dec_gemm' in r['Kernel_Name'] or 'dec_attention' in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dec = dec[-200 * n:]
pos = collections.defaultdict(list)
for i, r in enumerate(dec):
    pos[i % n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for i in range(n):
    d = sorted(pos[i]); tot += d[len(d) // 2]
    print('%2d %-28s grid=%sx%s med %.2f us min %.2f' % (i, dec[i]['Kernel_Name'].split('(')[0][-28:], dec[i]['Grid_Size_X'], dec[i]['Grid_Size_Y'], d[len(d) // 2] / 1e3, d[0] / 1e3))
print('sum of medians %.1f us; span/step %.1f us' % (tot / 1e3, (int(dec[-1]['End_Timestamp']) - int(dec[0]['Start_Timestamp'])) / 200e3))
