#!/bin/bash
# Round 4: FETCH_SIZE / WRITE_SIZE for the sparse model at BASELINE configs[4]'s shard shape; the counter collection is restricted to the
# evaluation and generator kernels (rocprofv3 died inside the tool on this workload in round 3 and in tools/r4_pmc_pass.sh when every
# kernel was instrumented).
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_pmc_sparse; mkdir -p $O
C4="--no-cpu --sparse --genes 50000 --samples 12500 --steps 4 --warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1500 rocprofv3 --pmc $C --kernel-include-regex "eval_sparse_kernel|gen_kernel" --output-format csv -d /tmp/pmcs_$C -- python $R/bench.py $C4 > $O/bench_sparse_c4shape_$C.json 2> $O/sparse_$C.err ) 2>&1 | grep real
  tail -2 $O/sparse_$C.err | cut -c1-200
done
python $R/tools/pmc_traffic.py /tmp/pmcs_FETCH_SIZE /tmp/pmcs_WRITE_SIZE $O/bench_sparse_c4shape_FETCH_SIZE.json $O/sparse_c4shape_pmc_traffic.json
