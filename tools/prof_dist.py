"""Per-kernel duration percentiles of a rocprofv3 --kernel-trace run: prof_dist.py <dir> [--window-ms X | --bench-json file]
--window-ms X (or the bench line's config.timed_region_ms_rank0): only the dispatches that ended within the last X ms of the trace --
the bench's timed region (nothing is launched behind it), so that the figures can be laid beside the line's own avg_launch_us."""
import csv, glob, json, sys, collections
args = sys.argv[1:]
win = None
if '--window-ms' in args: i = args.index('--window-ms'); win = float(args[i + 1]); del args[i:i + 2]
if '--bench-json' in args:
    i = args.index('--bench-json'); win = json.loads(open(args[i + 1]).read().strip().splitlines()[-1])['config']['timed_region_ms_rank0']; del args[i:i + 2]
recs = []
for f in glob.glob(args[0] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        recs.append((r['Kernel_Name'].split('(')[0][:48], int(r['Start_Timestamp']), int(r['End_Timestamp'])))
if win is not None and recs:
    t_last = max(e for _, _, e in recs)
    recs = [x for x in recs if x[2] >= t_last - win * 1e6]
    print('dispatches that ended within the last %.1f ms of the trace (the bench line\'s timed region): %d; their summed duration %.1f ms' % (win, len(recs), sum(e - s for _, s, e in recs) / 1e6))
rows = collections.defaultdict(list)
for k, s, e in recs: rows[k].append((e - s) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20: continue
    v.sort(); n = len(v)
    print('%-50s n %7d min %.2f p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f p95 %.2f p99 %.2f max %.2f mean %.2f total %.1f ms' % (k, n, v[0], v[n//10], v[n//4], v[n//2], v[3*n//4], v[9*n//10], v[95*n//100], v[99*n//100], v[-1], sum(v)/n, sum(v)/1e3))
