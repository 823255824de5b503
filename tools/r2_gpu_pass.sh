#!/bin/bash
# Round-2 GPU pass (run through gpurun): tests, smoke, bench at the driver's flags and at the defaults, rocprofv3 kernel trace of the same commands.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-pass}
O=gpurun_out/r2_$TAG; rm -rf $O; mkdir -p $O
if [ "$2" != "notests" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_s20_w5.json 2> $O/bench_s20_w5.err; tail -c 1500 $O/bench_s20_w5.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
( cd /tmp && rm -rf /tmp/prof && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $OLDPWD/bench.py --no-cpu --steps 20 --warmup 5 > $OLDPWD/$O/bench_s20_w5_under_rocprofv3.json 2> $OLDPWD/$O/rocprof.err )
python tools/prof_summary.py /tmp/prof > $O/rocprofv3_kernel_trace_summary_s20_w5.txt 2>&1
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats_s20_w5.csv 2>/dev/null
head -8 $O/rocprofv3_kernel_trace_summary_s20_w5.txt
