"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, workgroup size) count / mean / median / p90 / total."""
import csv, glob, sys, collections, statistics
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:60]
        wg = r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '?'
        grid = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
        rows[(name, wg, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print('%-62s %6s %8s %8s %9s %9s %9s %10s %6s' % ('kernel', 'wg', 'grid', 'count', 'mean_us', 'med_us', 'p90_us', 'total_ms', 'pct'))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-62s %6s %8s %8d %9.2f %9.2f %9.2f %10.2f %6.1f' % (k[0], k[1], k[2], len(v), statistics.mean(v), v[len(v)//2], v[int(len(v)*0.9)], sum(v)/1e3, 100*sum(v)/tot))
