import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'].split('(')[0][:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in rows.items():
    if len(v) < 100: continue
    v.sort(); n = len(v)
    print('%-42s n %7d min %.2f p1 %.2f p5 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f max %.2f mean %.2f' % (k, n, v[0], v[n//100], v[n//20], v[n//4], v[n//2], v[3*n//4], v[95*n//100], v[-1], sum(v)/n))
