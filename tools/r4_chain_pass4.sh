#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="${CONFIGS:-ab_libs/chain10.so;ab_libs/chain9.so}" TAG=${TAG:-chain10}
bash tools/r4_chain_ab.sh
