#!/bin/bash
# One parameterised GPU pass (through gpurun), replacing the per-round r2_* / r3_* / r4_* scripts:
#   TAG=name [TESTS="-k expr" | TESTS=all] [BENCH="args;args"] [AB="libA.so[ ENV=1];libB.so"] [PROF="bench args"] [SPARSE=1] bash tools/gpu_pass.sh
# TESTS   pytest -m gpu selection ("all": the whole suite + smoke())
# BENCH   ';'-separated bench.py argument lists, each run once (JSON lines -> bench_<i>.json)
# AB      ';'-separated "library [ENV=VAL ...]" configurations, alternating --no-cpu runs, two rounds (tools/dev_ab_env.sh)
# PROF    rocprofv3 --kernel-trace --stats of `bench.py <PROF>`: kernel stats, trace summary, duration percentiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/${TAG:-pass}; rm -rf $O; mkdir -p $O
if [ "$TESTS" = "all" ]; then
  timeout ${TEST_TIMEOUT:-3000} python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
elif [ -n "$TESTS" ]; then
  eval timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q $TESTS > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
fi
if [ -n "$BENCH" ]; then
  IFS=';' read -ra BL <<< "$BENCH"; i=0
  for B in "${BL[@]}"; do
    timeout 900 python bench.py $B > $O/bench_$i.json 2> $O/bench_$i.err; python tools/bench_brief.py $O/bench_$i.json "$B"; i=$((i+1))
  done
fi
if [ -n "$AB" ]; then
  IFS=';' read -ra CFG <<< "$AB"
  BENCH_ARGS="${AB_ARGS:---steps 20 --warmup 5}" bash tools/dev_ab_env.sh "${CFG[@]}" -- > $O/ab.txt 2>&1; cat $O/ab.txt
fi
if [ -n "$PROF" ]; then
  ( cd /tmp && rm -rf /tmp/prof && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu $PROF > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
  python tools/prof_summary.py /tmp/prof > $O/rocprofv3_kernel_trace_summary.txt 2>&1; head -8 $O/rocprofv3_kernel_trace_summary.txt
  cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
  python tools/prof_dist.py /tmp/prof > $O/rocprofv3_kernel_duration_percentiles.txt 2>&1; cat $O/rocprofv3_kernel_duration_percentiles.txt
  python tools/prof_dist.py /tmp/prof --bench-json $O/bench_under_rocprofv3.json > $O/rocprofv3_kernel_duration_percentiles_timed_window.txt 2>&1; cat $O/rocprofv3_kernel_duration_percentiles_timed_window.txt
fi
