#!/bin/bash
# Round 4's closing measurement pass (through gpurun): GPU tests, the bench at the driver's flags and at the defaults, rocprofv3 kernel trace + stats of the
# driver's command, the sparse model (C3-sized product and BASELINE configs[4]'s shard shape) with its kernel trace, the batched 8-chain mode.  The counter (--pmc) passes are tools/r4_pmc_pass.sh.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_s20_w5.json 2> $O/bench_s20_w5.err; cut -c1-300 $O/bench_s20_w5.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
( cd /tmp && rm -rf /tmp/prof && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_s20_w5_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_summary.py /tmp/prof > $O/rocprofv3_kernel_trace_summary.txt 2>&1
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
head -8 $O/rocprofv3_kernel_trace_summary.txt
python tools/prof_dist.py /tmp/prof > $O/rocprofv3_kernel_duration_percentiles.txt 2>&1
timeout 900 python bench.py --no-cpu --sparse --steps 20 --warmup 5 > $O/bench_sparse.json 2>/dev/null; cut -c1-200 $O/bench_sparse.json
C4="--no-cpu --sparse --genes 50000 --samples 12500 --steps 20 --warmup 5"
timeout 1500 python bench.py $C4 > $O/bench_sparse_c4shape.json 2> $O/bench_sparse_c4shape.err; cut -c1-300 $O/bench_sparse_c4shape.json
( cd /tmp && rm -rf /tmp/prof4 && timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python $R/bench.py $C4 > $O/bench_sparse_c4shape_under_rocprofv3.json 2> $O/rocprof4.err )
python tools/prof_summary.py /tmp/prof4 > $O/sparse_c4shape_rocprofv3_kernel_trace_summary.txt 2>&1
cp $(find /tmp/prof4 -name '*kernel_stats.csv' | head -1) $O/sparse_c4shape_rocprofv3_kernel_stats.csv 2>/dev/null
head -6 $O/sparse_c4shape_rocprofv3_kernel_trace_summary.txt
timeout 600 python bench.py --no-cpu --chains 8 --steps 20 --warmup 5 > $O/bench_chains8.json 2>/dev/null; cut -c1-200 $O/bench_chains8.json
