"""Timeline analysis of a rocprofv3 --kernel-trace CSV: kernel time vs gaps between consecutive dispatches."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:28], r.get('Workgroup_Size_X') or r.get('Workgroup_Size')))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for i in range(len(rows) - 1):
    a, b = rows[i], rows[i + 1]
    gap[(a[2] + '/' + str(a[3]), b[2] + '/' + str(b[3]))].append((b[0] - a[1]) / 1e3)
    dur[a[2] + '/' + str(a[3])].append((a[1] - a[0]) / 1e3)
span = (rows[-1][1] - rows[0][0]) / 1e6
ktot = sum(sum(v) for v in dur.values()) / 1e3
print('span %.1f ms, kernel time %.1f ms (%.1f%%), dispatches %d' % (span, ktot, 100 * ktot / span, len(rows)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v.sort(); print('  kernel %-36s n=%7d mean %7.2f med %7.2f p90 %7.2f us total %8.1f ms' % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)], sum(v) / 1e3))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v.sort(); print('  gap %-58s n=%7d mean %7.2f med %7.2f p90 %7.2f us total %8.1f ms' % (k[0] + ' -> ' + k[1], len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)], sum(v) / 1e3))
