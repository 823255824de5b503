#!/bin/bash
# HBM-traffic counters for the batched multi-chain bench (separate --pmc passes, no trace domains, no graph replays): tools/pmc_pass_chains.sh C G
# (ROCm 7.2: rocprofv3's counter collection dies with SIGSEGV inside the tool ~100 s into this workload, with 8 and with 32 chains -- no
# counter figures for the batched launches this round; the kernel trace of the same command works, tools/r2_gpu_chain_trace.sh)
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; C=${1:-32}; G=${2:-1}; O=$R/gpurun_out/r2_pmc_chains$C; mkdir -p $O
for K in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$K
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1500 rocprofv3 --pmc $K --output-format csv -d /tmp/pmcc_$K -- python $R/bench.py --no-cpu --chains $C --chain-groups $G --steps 6 --warmup 2 > $O/bench_$K.json 2> $O/$K.err ) 2>&1 | grep real
done
python $R/tools/pmc_summary.py /tmp/pmcc_FETCH_SIZE /tmp/pmcc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -8 $O/pmc_summary.txt
