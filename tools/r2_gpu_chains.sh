#!/bin/bash
# batched multi-chain launches: parity tests, then bench --chains in the batched mode with 1 / 2 / 4 groups and in the threads mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2_${1:-c}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "batched or sparse or gwcogaps or sccogaps" --durations=8 > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_single_s20.json 2>/dev/null; cut -c1-330 $O/bench_single_s20.json
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 --sparse > $O/bench_sparse_s20.json 2>/dev/null; python - <<PY
import json; d=json.load(open("$O/bench_sparse_s20.json")); print("sparse", d["value"], [(k["kernel"][:24], k["sampler"], round(k["avg_launch_us"],2)) for k in d["roofline"]["kernels"]])
PY
for spec in "8 1" "8 2" "8 4" "4 2" "16 4"; do set -- $spec; timeout 600 python bench.py --no-cpu --chains $1 --chain-groups $2 --steps 20 --warmup 5 > $O/bench_chains$1_groups$2.json 2>/dev/null; python - <<PY
import json; d=json.load(open("$O/bench_chains$1_groups$2.json")); r=d["roofline"]; print("chains $1 groups $2: %.2f M/s  ms/step %.1f  kt/wall %.2f  alg GB/s over wall %.0f" % (d["value"]/1e6, d["ms_per_step"], r["sampled_kernel_time_over_wall"], d["config"]["algorithmic_GBps_over_wall"]), [(k["kernel"][8:22], round(k["avg_launch_us"],1)) for k in r["kernels"][:4]])
PY
done
