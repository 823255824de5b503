#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain13.so;ab_libs/chain11.so" TAG=chain13
export TESTS="-k 'headline_shape_stepwise'"
bash tools/r4_chain_ab.sh
