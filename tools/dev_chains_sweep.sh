#!/bin/bash
# dev: batched multi-chain bench over (chains, groups) pairs: tools/dev_chains_sweep.sh "16 1" "16 2" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2_sweep; mkdir -p $O
for spec in "$@"; do set -- $spec; timeout 900 python bench.py --no-cpu --chains $1 --chain-groups $2 --steps 20 --warmup 5 > $O/bench_chains$1_groups$2.json 2>$O/err_$1_$2.txt; python - <<PY
import json
try:
    d=json.load(open("$O/bench_chains$1_groups$2.json")); r=d["roofline"]; print("chains $1 groups $2: %.2f M/s  ms/step %.1f  path frac %.3f" % (d["value"]/1e6, d["ms_per_step"], r["frac"]), [(k["kernel"].split()[1][:4]+k["kernel"].split("sampler ")[1][:1], round(k["avg_launch_us"],1)) for k in r["kernels"][:4]])
except Exception as e: print("chains $1 groups $2 failed", e, open("$O/err_$1_$2.txt").read()[-400:])
PY
done
