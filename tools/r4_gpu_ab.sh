#!/bin/bash
# Round 4 (through gpurun): [GPU suite] + A/B of (library, environment) pairs + [rocprofv3 duration percentiles of the current build].
#   CONFIGS="ab_libs/a.so;ab_libs/b.so ENV=1" TESTS=1 PROF=1 TAG=pass4 bash tools/r4_gpu_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_${TAG:-ab}; rm -rf $O; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log; fi
IFS=';' read -ra CFG <<< "$CONFIGS"
bash tools/dev_ab_env.sh "${CFG[@]}" -- > $O/ab.txt 2>&1; cat $O/ab.txt
if [ -n "$PROF" ]; then
  ( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
  python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
fi
