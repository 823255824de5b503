#!/bin/bash
# kernel trace of the batched multi-chain bench: kernel durations and the idle time between launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2_${1:-chaintrace}; rm -rf $O; mkdir -p $O
C=${2:-8}; G=${3:-1}
( cd /tmp && rm -rf /tmp/profc && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/profc -- python $OLDPWD/bench.py --no-cpu --chains $C --chain-groups $G --steps 6 --warmup 2 > $O/bench.json 2> $O/rocprof.err )
python tools/prof_summary.py /tmp/profc > $O/summary.txt 2>&1; head -12 $O/summary.txt
python tools/prof_gaps.py /tmp/profc > $O/gaps.txt 2>&1; head -24 $O/gaps.txt
