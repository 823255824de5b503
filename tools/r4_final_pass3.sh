#!/bin/bash
# GPU suite + the closing measurement pass in one call (final build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/r4_chain_gpu_suite.sh
bash tools/r4_final_pass2.sh
