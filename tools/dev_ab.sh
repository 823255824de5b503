#!/bin/bash
# dev: A/B of two builds of the library on the GPU box: tools/dev_ab.sh <libA> <libB> [bench args]; alternates runs
A=$1; B=$2; shift 2
cd /root/repo
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for i in 1 2; do
  for L in $A $B; do
    cp $L cogaps_amd/csrc/libcogaps_hip.so
    timeout 300 python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'evalA %.2f evalP %.2f gen_ms %.0f' % (d['roofline']['avg_launch_us'], d['roofline']['other_sampler']['avg_batch_us'], d['config']['gen_kernel_ms_rank0']))"
  done
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
