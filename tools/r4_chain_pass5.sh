#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain12.so;ab_libs/chain11.so" TAG=chain12
export TESTS="-k 'headline_shape_stepwise'"
bash tools/r4_chain_ab.sh
