#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain23.so;ab_libs/chain20.so" TAG=chain23
export TESTS="-k 'headline_shape_stepwise or gist_stepwise or benchmarked_chain_end_to_end'"
bash tools/r4_chain_ab.sh
