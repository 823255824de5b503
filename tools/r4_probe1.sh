#!/bin/bash
# Round 4, probe 1 (through gpurun): this box's baseline, the potential of a scalar step without table reads (FAKELUT build: timing
# only), the timeline of a two-round generator launch, the duration distribution of the generator under rocprofv3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_probe1; rm -rf $O; mkdir -p $O
C=cogaps_amd/csrc
cp $C/libcogaps_hip.so /tmp/base.so
bash tools/dev_ab.sh "/tmp/base.so $C/libcogaps_hip_FAKELUT_DEV.so" --steps 20 --warmup 5 > $O/ab_fakelut.txt 2>&1; cat $O/ab_fakelut.txt
timeout 300 python tools/dev_timeline_round2.py 60 > $O/timeline_round2.txt 2>&1; tail -80 $O/timeline_round2.txt
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
