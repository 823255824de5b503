#!/bin/bash
# A/B of (library build, environment) pairs on the GPU box, alternating, two rounds (libraries live beside the product library as
# cogaps_amd/csrc/libcogaps_hip_AB_<name>.so -- untracked, removed after the pass):
#   tools/dev_ab_env.sh "cogaps_amd/csrc/libcogaps_hip_AB_a.so" "cogaps_amd/csrc/libcogaps_hip_AB_a.so COGAPS_NO_CHAIN=1" -- --steps 20 --warmup 5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFG=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFG+=("$1"); shift; done; shift
cp cogaps_amd/csrc/libcogaps_hip.so /tmp/keep.so
for i in $(seq 1 ${AB_ROUNDS:-2}); do
  for C in "${CFG[@]}"; do
    set -- $C; L=$1; shift
    cp $L cogaps_amd/csrc/libcogaps_hip.so
    env "$@" timeout 300 python bench.py --no-cpu ${BENCH_ARGS:---steps 20 --warmup 5} 2>/dev/null | tail -1 > /tmp/ab_line.json
    python tools/bench_brief.py /tmp/ab_line.json "$(echo $C | sed 's#cogaps_amd/csrc/libcogaps_hip_AB_##')"
  done
done
cp /tmp/keep.so cogaps_amd/csrc/libcogaps_hip.so
