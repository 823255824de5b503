#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain21_w320.so;ab_libs/chain21_w256.so;ab_libs/chain20.so" TAG=chain21
export TESTS="-k 'headline_shape_stepwise or gist_stepwise'"
bash tools/r4_chain_ab.sh
