#!/bin/bash
# Round 4, pass 2 (through gpurun): the GPU suite on the current build, then an A/B of library builds (ab_libs/*.so named in $LIBS,
# alternating bench runs on this one box), then the generator's duration distribution under rocprofv3 for the current build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_pass2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
bash tools/dev_ab.sh "${LIBS:-ab_libs/base_r3.so ab_libs/lds_rounds.so}" --steps 20 --warmup 5 > $O/ab.txt 2>&1; cat $O/ab.txt
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
