#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4_chain_tl
export CONFIGS="${CONFIGS:-ab_libs/chain8.so;ab_libs/chain6.so}" TAG=${TAG:-chain8}
export TESTS="-k 'headline_shape_stepwise or gist_stepwise'"
bash tools/r4_chain_ab.sh
timeout 250 python tools/dev_chain_timeline.py 40 > gpurun_out/r4_chain_tl/chain_rt_$TAG.txt 2>&1; cat gpurun_out/r4_chain_tl/chain_rt_$TAG.txt
timeout 250 python tools/dev_timeline.py 40 > gpurun_out/r4_chain_tl/timeline_$TAG.txt 2>&1; head -46 gpurun_out/r4_chain_tl/timeline_$TAG.txt
