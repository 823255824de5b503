#!/bin/bash
# Round 4's closing pass with the chained launch (through gpurun): the bench at the driver's flags and the defaults, rocprofv3 kernel trace + stats + duration
# percentiles of the driver's command, the A/B against two launches per batch (COGAPS_NO_CHAIN=1) on the same box, the counter (--pmc) passes
# of the dense headline, one sparse line and the 8-chain line for the record.  (GPU tests: tools/r4_chain_gpu_suite.sh.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_final2; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_s20_w5.json 2> $O/bench_s20_w5.err; cut -c1-300 $O/bench_s20_w5.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
for i in 1 2; do for E in "" "COGAPS_NO_CHAIN=1"; do env $E timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-20s %8d  A %.2f + %.2f us  P %.2f + %.2f us  kt/wall %.3f' % ('$E' or 'chained launch', round(d['value']), k[0]['avg_launch_us'], k[2]['avg_launch_us'], k[1]['avg_launch_us'], k[3]['avg_launch_us'], d['roofline']['kernel_time_over_wall']))"; done; done | tee $O/ab_chained_vs_two_launches.txt
( cd /tmp && rm -rf /tmp/prof && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_s20_w5_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_summary.py /tmp/prof > $O/rocprofv3_kernel_trace_summary.txt 2>&1
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
head -8 $O/rocprofv3_kernel_trace_summary.txt
python tools/prof_dist.py /tmp/prof > $O/rocprofv3_kernel_duration_percentiles.txt 2>&1; cat $O/rocprofv3_kernel_duration_percentiles.txt
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( time COGAPS_NO_GRAPH=1 timeout -k 5 1200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/bench.py --no-cpu > $O/bench_$C.json 2> $O/$C.err ) 2>&1 | grep real
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/bench_FETCH_SIZE.json $O/pmc_traffic.json
python $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
cd $R
timeout 900 python bench.py --no-cpu --sparse --genes 50000 --samples 12500 --steps 20 --warmup 5 > $O/bench_sparse_c4shape.json 2>/dev/null; cut -c1-200 $O/bench_sparse_c4shape.json
timeout 600 python bench.py --no-cpu --chains 8 --steps 20 --warmup 5 > $O/bench_chains8.json 2>/dev/null; cut -c1-200 $O/bench_chains8.json
