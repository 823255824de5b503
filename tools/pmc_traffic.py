"""Per-launch HBM traffic of the evaluation kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes: the TCC block cannot hold both) over `python bench.py --no-cpu` with COGAPS_NO_GRAPH=1 (counter collection
hangs on replayed graphs).  For each kernel: the launches of the timed region = the last `launches` dispatches that
moved data (empty-queue launches, FETCH < 64 KB, are left out as bench.py leaves them out).  FETCH_SIZE is doubled
(gfx950 tallies the 128-byte requests of 16 B/lane coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is reported as the tool gives it (uncalibrated).  usage: pmc_traffic.py <fetch_dir> <write_dir> <bench_json> <out_json>"""
import csv, glob, json, sys, collections


def per_kernel(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                name = r['Kernel_Name'].split('(')[0]
                if 'gen_kernel<' in name or 'gen_apply_kernel<' in name: name = 'gen_kernel'          # (both window instantiations, with or without the update workgroups: one generator launch per batch)
                if 'chain_kernel<' in name: name = 'chain_kernel'      # (chained launch: evaluation of batch n + generator of batch n+1, one launch per batch)
                rows[name].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for v in rows.values():
        v.sort()
    return rows


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
bench = json.load(open(sys.argv[3]))
ks = bench['roofline']['kernels']        # [evaluation A, evaluation P, generator A, generator P, sync]
want = {'eval_kernel<0>': ks[0]['launches'], 'eval_kernel<1>': ks[1]['launches'], 'eval_kernel<2>': ks[1]['launches'], 'eval_kernel<4>': ks[1]['launches'],
        'gen_kernel': ks[2]['launches'] + ks[3]['launches']}      # (<1> + <2>: the two-launch split form of rounds 1-3; <4>: the one-launch form, whose updates are counted with the generator launch)
for i_, w_ in ((0, 'A'), (1, 'P')):
    if ks[i_]['kernel'].startswith('chain_kernel'): want['chain_kernel'] = want.get('chain_kernel', 0) + ks[i_]['launches']
if 'sparse' in bench['config']['workload']:      # the sparse model: one evaluation kernel per sampler (the wide form where a vector has more flag words than word owners),
    # or -- round 5 -- the chained launch (chain_sparse_kernel<WIN, WIDE>: evaluation + generator in one launch per batch)
    names = sorted({k for k in fetch if 'eval_sparse_kernel' in k or 'chain_sparse_kernel' in k})
    want = {'gen_kernel': ks[2]['launches'] + ks[3]['launches']} if (ks[2]['launches'] + ks[3]['launches']) else {}
    for k in names:
        want[k] = 0       # all its launches that moved data
    # (the run walks through the whole schedule before its timed steps: the populated chain = the last quarter of each kernel's launches)
    quarter = True
out = {}
for name, n in want.items():
    fk = [k for k in fetch if (name == k or (name in k and 'sparse' not in name))]; wk = [k for k in write if (name == k or (name in k and 'sparse' not in name))]
    if not fk or not wk:
        continue
    f, w = fetch[fk[0]], write[wk[0]]
    n = n or (max(1, len(f) // 4) if 'quarter' in globals() else len(f))
    # the same dispatches in both passes (deterministic run): select on the fetch pass, by position
    idx = [i for i, (_, v) in enumerate(f) if v >= 64.0 or name.startswith('gen') or name.startswith('chain')][-n:]      # (generator: the last n launches, empty-queue ones included)
    fb = sum(f[i][1] for i in idx) / len(idx) * 1024.0 * 2.0
    wb = sum(w[i][1] for i in idx if i < len(w)) / len(idx) * 1024.0
    out[name] = {'launches': len(idx), 'fetch_bytes_per_launch_corrected_x2': fb, 'write_bytes_per_launch': wb, 'hbm_bytes_per_launch': fb + wb,
                 'all_launches_in_run': len(f)}
    print(name, out[name])
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- COGAPS_NO_GRAPH=1 python bench.py --no-cpu', 'kernels': out,
           # the build the counters were collected on (cogaps_source_hash) and the workload: bench.py quotes the figures only for this build and shape
           'lib_source_hash': bench['roofline'].get('lib_source_hash'), 'workload': bench['config']['workload'],
           # the bench line's own algorithmic figures, under the names the line gives its kernels (no hand-typed kernel names here)
           'bench_algorithmic_bytes_per_launch': dict([('%s [%s]' % (k['kernel'].split(' (')[0], k['sampler']), k['bytes_per_launch']) for k in ks[:2]] + [('path (per batch)', bench['roofline']['bytes_per_launch'])])},
          open(sys.argv[4], 'w'), indent=1)
