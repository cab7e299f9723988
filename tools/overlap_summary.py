"""Concurrency view of a rocprofv3 kernel_trace.csv: busy union, idle time, mean kernels in flight, per-queue counts,
per-kernel duration sums over the last `frac` of the trace (steady state)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-36:], r['Queue_Id']) for r in rows))
t_lo = ev[0][0]; t_hi = max(e[1] for e in ev)
cut = t_lo + (t_hi - t_lo) * float(sys.argv[2]) if len(sys.argv) > 2 else t_lo
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
# union
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _, _ in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
print('span %.2f ms  busy-union %.2f ms (%.0f%%)  sum-of-durations %.2f ms  mean in flight %.2f' % (span / 1e6, busy / 1e6, 100 * busy / span, tot / 1e6, tot / span))
q = collections.Counter(e[3] for e in ev); print('dispatches per queue', dict(q))
d = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in ev: d[n][0] += 1; d[n][1] += e - s
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:8]:
    print('  %-38s %7d calls  %8.2f ms  avg %7.2f us' % (n, c, t / 1e6, t / c / 1e3))
