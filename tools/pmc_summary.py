"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, launches and the mean of each counter per launch.
FETCH_SIZE / WRITE_SIZE are in kilobytes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so `fetch_bytes_corrected` = FETCH_SIZE * 1024 * 2."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0][:70]
            rows[name][r['Counter_Name']].append((int(r.get('Dispatch_Id', 0) or 0), float(r['Counter_Value'])))
print('%-72s %8s  %s' % ('kernel', 'launches', 'mean per launch'))
for name, cs in sorted(rows.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
    n = max(len(v) for v in cs.values())
    parts = []
    for c, v in sorted(cs.items()):
        v = [x for _, x in sorted(v)]
        m = sum(v) / len(v)
        tail = v[-max(1, len(v) // 4):]; mt = sum(tail) / len(tail)       # the run's last quarter of launches: the populated chain
        if c == 'FETCH_SIZE': parts.append('FETCH_SIZE %.1f KB (corrected x2: %.3f MB; last quarter of the launches %.3f MB)' % (m, m * 2048 / 1e6, mt * 2048 / 1e6))
        elif c == 'WRITE_SIZE': parts.append('WRITE_SIZE %.1f KB (%.3f MB; last quarter %.3f MB)' % (m, m * 1024 / 1e6, mt * 1024 / 1e6))
        else: parts.append('%s %.1f' % (c, m))
    print('%-72s %8d  %s' % (name, n, '; '.join(parts)))
