#!/bin/bash
# the sparse model at BASELINE configs[4]'s shard shape (A batches of ~250 proposals: half of them outgrow a 256-attempt window) by generator window
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for L in ab_libs/chain13.so ab_libs/chain_w384.so ab_libs/chain13.so ab_libs/chain_w384.so; do
  cp $L cogaps_amd/csrc/libcogaps_hip.so
  timeout 600 python bench.py --no-cpu --sparse --genes 50000 --samples 12500 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-24s %8d  A eval %.2f gen %.2f  P eval %.2f gen %.2f  avgq %.0f / %.0f' % ('$L'.replace('ab_libs/',''), round(d['value']), k[0]['avg_launch_us'], k[2]['avg_launch_us'], k[1]['avg_launch_us'], k[3]['avg_launch_us'], d['config']['avg_queue_A'], d['config']['avg_queue_P']))"
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
