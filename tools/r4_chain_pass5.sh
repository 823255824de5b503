#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain13.so;ab_libs/chain_g248.so;ab_libs/chain_g255.so" TAG=chaingrid
export TESTS="-k 'headline_shape_stepwise'"
bash tools/r4_chain_ab.sh
