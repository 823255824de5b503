#!/bin/bash
# rocprofv3 duration percentiles of the driver's command for a list of library builds: LIBS="ab_libs/a.so ab_libs/b.so" TAG=x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_${TAG:-prof}; mkdir -p $O
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for L in $LIBS; do
  cp $L cogaps_amd/csrc/libcogaps_hip.so
  N=$(basename $L .so)
  ( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3_$N.json 2> $O/rocprof_$N.err )
  echo "== $N"; python tools/prof_dist.py /tmp/prof 2>&1 | grep -E "chain_kernel|gen_|eval_kernel" | tee $O/percentiles_$N.txt
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
