#!/bin/bash
# Round 4: HBM-traffic counters for the bench command (dense headline and the sparse model at BASELINE configs[4]'s shard shape), as the
# MI355X guide prescribes: separate --pmc passes, no trace domains, plain launches (COGAPS_NO_GRAPH=1: counter collection hangs on
# replayed graphs).  The records carry the library's source hash; bench.py quotes them only for that build.
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_pmc; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/bench.py --no-cpu > $O/bench_$C.json 2> $O/$C.err ) 2>&1 | grep real
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/bench_FETCH_SIZE.json $O/pmc_traffic.json
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
C4="--no-cpu --sparse --genes 50000 --samples 12500 --steps 4 --warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1500 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcs_$C -- python $R/bench.py $C4 > $O/bench_sparse_c4shape_$C.json 2> $O/sparse_$C.err ) 2>&1 | grep real
  tail -2 $O/sparse_$C.err
done
python $R/tools/pmc_traffic.py /tmp/pmcs_FETCH_SIZE /tmp/pmcs_WRITE_SIZE $O/bench_sparse_c4shape_FETCH_SIZE.json $O/sparse_c4shape_pmc_traffic.json
