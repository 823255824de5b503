#!/bin/bash
# The round's closing GPU pass (run through gpurun): tests, smoke, default bench, rocprofv3 kernel trace of the same command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --no-cpu > $OLDPWD/$O/bench_under_rocprofv3.json 2> $OLDPWD/$O/rocprof.err )
python tools/prof_summary.py /tmp/prof > $O/rocprofv3_kernel_trace_summary.txt 2>&1
python tools/prof_dist.py /tmp/prof > $O/rocprofv3_kernel_duration_percentiles.txt 2>&1
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
head -6 $O/rocprofv3_kernel_trace_summary.txt
timeout 300 python bench.py --no-cpu --chains 2 > $O/bench_chains2.json 2>/dev/null; cut -c1-250 $O/bench_chains2.json
timeout 300 python bench.py --no-cpu --chains 3 > $O/bench_chains3.json 2>/dev/null; cut -c1-250 $O/bench_chains3.json
timeout 300 python bench.py --no-cpu --chains 4 > $O/bench_chains4.json 2>/dev/null; cut -c1-250 $O/bench_chains4.json
timeout 600 python bench.py --no-cpu --sparse > $O/bench_sparse.json 2>/dev/null; cut -c1-250 $O/bench_sparse.json
