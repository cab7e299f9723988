"""List the dispatches of the LAST repetition in a rocprofv3 kernel_trace.csv: name, grid, duration (us).
   python tools/trace_list.py kernel_trace.csv [name substring] [dispatches per repetition]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if sub in r['Kernel_Name']]
per = int(sys.argv[3]) if len(sys.argv) > 3 else len(sel)
tot = 0.0
for r in sel[-per:]:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += us
    print('%-60s grid %-14s wg %-5s %9.1f us' % (r['Kernel_Name'].split('(')[0][-60:], 'x'.join(str(r[k]) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z') if k in r) or r.get('Grid_Size', '?'), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), us))
print('total %.1f us over %d dispatches' % (tot, min(per, len(sel))))
