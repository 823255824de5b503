#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_chain_p2; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_chain.json 2>/dev/null
COGAPS_NO_CHAIN=1 timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_nochain.json 2>/dev/null
python - <<'PY'
import json
for n in ("chain", "nochain"):
    d = json.load(open("gpurun_out/r4_chain_p2/bench_%s.json" % n))
    print(n, round(d["value"]), d["ms_per_step"], d["config"]["empty_launch_us"], d["config"]["launches_per_batch"], d["config"]["batches_rank0"])
PY
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/rocprof.err )
python tools/prof_dist.py /tmp/prof > $O/kernel_duration_percentiles.txt 2>&1; cat $O/kernel_duration_percentiles.txt
python tools/prof_gaps.py /tmp/prof > $O/kernel_gaps.txt 2>&1; head -30 $O/kernel_gaps.txt
