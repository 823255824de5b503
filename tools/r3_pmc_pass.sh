#!/bin/bash
# HBM-traffic counters for the bench command, as the MI355X guide prescribes: separate --pmc passes, no trace domains.
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3_pmc; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/bench.py --no-cpu > $O/bench_$C.json 2> $O/$C.err ) 2>&1 | grep real
  ls -la $(find /tmp/pmc_$C -name '*counter_collection.csv') | awk '{print $5, $9}'
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/bench_FETCH_SIZE.json $O/pmc_traffic.json
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
# the batched multi-chain launches (8 chains, one batch, plain launches, few steps): rocprofv3's counter collection crashed on this
# workload in round 2; tried again, with the tool's own stderr kept
O8=$R/gpurun_out/r3_pmc_chains8; mkdir -p $O8
for K in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$K
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 900 rocprofv3 --pmc $K --output-format csv -d /tmp/pmcc_$K -- python $R/bench.py --no-cpu --chains 8 --chain-groups 1 --steps 4 --warmup 1 > $O8/bench_$K.json 2> $O8/$K.err ) 2>&1 | grep real
  echo "rc $?" ; tail -3 $O8/$K.err
done
python $R/tools/pmc_summary.py /tmp/pmcc_FETCH_SIZE /tmp/pmcc_WRITE_SIZE > $O8/pmc_summary.txt 2>&1; head -8 $O8/pmc_summary.txt
