#!/bin/bash
# dev: A/B of library builds under the batched multi-chain bench: tools/dev_ab_chains.sh "<lib> ..." [bench args, e.g. --chains 8 --chain-groups 1]
LIBS=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for i in 1 2; do
  for L in $LIBS; do
    cp $L cogaps_amd/csrc/libcogaps_hip.so
    timeout 400 python bench.py --no-cpu --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-20s %-28s %9d  ' % ('$L'.split('/')[-1], '$*', round(d['value'])) + ' '.join('%s %.1f' % (x['kernel'].split()[1][:4] + x['kernel'].split('sampler ')[1][:1], x['avg_launch_us']) for x in k[:4]))"
  done
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
