#!/bin/bash
# Round 4, chained launch (csrc/chain_kernel.h): quick GPU parity subset, then the A/B of the chained form against two launches per batch
# (COGAPS_NO_CHAIN=1), [rocprofv3 duration percentiles].   TESTS="-k 'expr'" PROF=1 TAG=x CONFIGS="a.so;a.so COGAPS_NO_CHAIN=1" bash tools/r4_chain_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_${TAG:-chain}; rm -rf $O; mkdir -p $O
if [ -n "$TESTS" ]; then eval timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q $TESTS > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log; fi
IFS=';' read -ra CFG <<< "$CONFIGS"
if [ ${#CFG[@]} -gt 0 ]; then bash tools/dev_ab_env.sh "${CFG[@]}" -- > $O/ab.txt 2>&1; cat $O/ab.txt; fi
if [ -n "$PROF" ]; then
  ( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
  python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
fi
