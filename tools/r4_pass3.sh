#!/bin/bash
# Round 4, pass 3 (through gpurun): GPU suite on the current build; A/B of the round's two changes on one box (later rounds of a batch in
# the LDS table: cur vs cur_lds1; the split evaluation as one launch with its updates beside the next generator launch: cur vs the same
# library with COGAPS_SPLIT_TWO_LAUNCHES=1; the number of update workgroups); timeline of a two-round generator launch; rocprofv3 durations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_pass3; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
bash tools/dev_ab_env.sh "ab_libs/cur_lds1.so COGAPS_SPLIT_TWO_LAUNCHES=1" "ab_libs/cur.so COGAPS_SPLIT_TWO_LAUNCHES=1" "ab_libs/cur.so" "ab_libs/cur.so COGAPS_APPLY_GRID=511" "ab_libs/cur.so COGAPS_APPLY_GRID=127" -- > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 300 python tools/dev_timeline_round2.py 60 > $O/timeline_round2.txt 2>&1; grep -A48 "^wave 3" $O/timeline_round2.txt | head -60
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
