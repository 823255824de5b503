#!/bin/bash
# dev: A/B of (library build, environment) pairs on the GPU box, alternating, two rounds:
#   tools/dev_ab_env.sh "ab_libs/a.so" "ab_libs/a.so COGAPS_SPLIT_TWO_LAUNCHES=1" "ab_libs/b.so" -- --steps 20 --warmup 5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFG=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFG+=("$1"); shift; done; shift
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for i in 1 2; do
  for C in "${CFG[@]}"; do
    set -- $C; L=$1; shift
    cp $L cogaps_amd/csrc/libcogaps_hip.so
    env "$@" timeout 300 python bench.py --no-cpu ${BENCH_ARGS:---steps 20 --warmup 5} 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-44s %8d  evalA %.2f evalP %.2f genA %.2f genP %.2f  kt/wall %.3f' % ('$C'.replace('ab_libs/','')[:44], round(d['value']), k[0]['avg_launch_us'], k[1]['avg_launch_us'], k[2]['avg_launch_us'], k[3]['avg_launch_us'], d['roofline']['kernel_time_over_wall']))"
  done
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
