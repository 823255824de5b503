#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain14.so;ab_libs/chain13.so" TAG=chain14
export TESTS="-k 'headline_shape_stepwise'"
bash tools/r4_chain_ab.sh
