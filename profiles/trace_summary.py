"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations and the gaps between consecutive dispatches."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-40:]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[n].append(e - s)
    if prev_end is not None:
        gap[n].append(s - prev_end)
    prev_end = e
print('%-42s %7s %9s %9s %9s | gap-before: %8s %8s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'median', 'avg'))
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = sorted(gap[n]) or [0]
    print('%-42s %7d %9.2f %9.2f %9.2f | %8.2f %8.2f' % (n, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3,
                                                         g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3))
