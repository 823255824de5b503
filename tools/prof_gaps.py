"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where a stream of launches is not busy.

usage: prof_gaps.py <rocprof output dir>
Prints, per (previous kernel -> next kernel) pair: count, mean / median / p90 gap (us), total gap (ms); and busy / idle totals."""
import csv, glob, sys, collections, statistics
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0][-28:]))
ev.sort()
gaps = collections.defaultdict(list)
busy = 0; idle = 0; end = None; prev = None
for s, e, n in ev:
    if end is not None:
        g = (s - end) / 1e3
        if g < 2000: gaps[(prev, n)].append(g)     # (longer pauses: host phases between updates, listed separately)
        else: gaps[('(pause > 2 ms)', n)].append(g)
        idle += max(0, s - end)
    busy += e - s
    end = max(end or 0, e); prev = n
print('kernels %d  busy %.1f ms  idle %.1f ms  busy fraction %.3f' % (len(ev), busy / 1e6, idle / 1e6, busy / max(1, busy + idle)))
print('%-30s -> %-30s %8s %9s %9s %9s %10s' % ('previous', 'next', 'count', 'mean_us', 'med_us', 'p90_us', 'total_ms'))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-30s -> %-30s %8d %9.2f %9.2f %9.2f %10.2f' % (k[0], k[1], len(v), statistics.mean(v), v[len(v) // 2], v[int(len(v) * 0.9)], sum(v) / 1e3))
