cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3_final2; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --no-cpu --sparse --steps 20 --warmup 5 > $O/bench_sparse.json 2>/dev/null; cut -c1-200 $O/bench_sparse.json
C4="--no-cpu --sparse --genes 50000 --samples 12500 --steps 20 --warmup 5"
timeout 1500 python bench.py $C4 > $O/bench_sparse_c4shape.json 2> $O/bench_sparse_c4shape.err; cut -c1-300 $O/bench_sparse_c4shape.json
( cd /tmp && rm -rf /tmp/prof4 && timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python $R/bench.py $C4 > $O/bench_sparse_c4shape_under_rocprofv3.json 2> $O/rocprof4.err )
python tools/prof_summary.py /tmp/prof4 > $O/sparse_c4shape_rocprofv3_kernel_trace_summary.txt 2>&1
cp $(find /tmp/prof4 -name '*kernel_stats.csv' | head -1) $O/sparse_c4shape_rocprofv3_kernel_stats.csv 2>/dev/null
head -7 $O/sparse_c4shape_rocprofv3_kernel_trace_summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sparse or configs4" > $O/gpu_sparse_tests.log 2>&1; tail -2 $O/gpu_sparse_tests.log
