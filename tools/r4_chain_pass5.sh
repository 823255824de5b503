#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="${CONFIGS:-ab_libs/chain19.so;ab_libs/chain18.so}" TAG=${TAG:-chain19}
export TESTS="-k headline_shape_stepwise"
export TEST_TIMEOUT=1500
bash tools/r4_chain_ab.sh
