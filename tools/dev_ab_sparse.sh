#!/bin/bash
# dev: A/B of library builds on the sparse bench: tools/dev_ab_sparse.sh "<lib> <lib> ..." [bench args]; alternates runs, two rounds
LIBS=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for i in 1 2; do
  for L in $LIBS; do
    cp $L cogaps_amd/csrc/libcogaps_hip.so
    timeout 600 python bench.py --no-cpu --sparse "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-28s %8d  evalA %.2f evalP %.2f genA %.2f genP %.2f' % ('$L'.split('/')[-1], round(d['value']), k[0]['avg_launch_us'], k[1]['avg_launch_us'], k[2]['avg_launch_us'], k[3]['avg_launch_us']))"
  done
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
