#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CONFIGS="ab_libs/chain22.so;ab_libs/chain20.so" TAG=chain22
export TESTS="-k 'headline_shape_stepwise or gist_stepwise'"
bash tools/r4_chain_ab.sh
timeout 600 python tools/dev_chain_log.py 60 2>&1 | grep -A8 "two rounds"
