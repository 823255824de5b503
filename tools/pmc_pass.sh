#!/bin/bash
# HBM-traffic counters for a bench command, as the MI355X guide prescribes: separate --pmc passes (one counter each), no trace domains,
# COGAPS_NO_GRAPH=1 (counter collection hangs on replayed graphs).
#   TAG=r05_pmc [ARGS="--sparse --genes 50000 --samples 12500 --steps 4 --warmup 1"] [KREGEX='eval_sparse|gen_kernel'] [COUNTERS="FETCH_SIZE WRITE_SIZE"] bash tools/pmc_pass.sh
# KREGEX restricts the instrumented kernels (rocprofv3 --kernel-include-regex): the tool has died instrumenting every kernel of the sparse model's run.
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-pmc}; mkdir -p $O
INC=(); [ -n "$KREGEX" ] && INC=(--kernel-include-regex "$KREGEX")
for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/pmc_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 ${PMC_TIMEOUT:-1200} rocprofv3 --pmc $C "${INC[@]}" --output-format csv -d /tmp/pmc_$C -- python $R/bench.py --no-cpu $ARGS > $O/bench_$C.json 2> $O/$C.err ) 2>&1 | grep real
  echo "$C rc=$? $(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1 | xargs -r ls -la | awk '{print $5}') bytes of counter records"; tail -2 $O/$C.err
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/bench_FETCH_SIZE.json $O/pmc_traffic.json
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -14 $O/pmc_summary.txt
