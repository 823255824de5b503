#!/bin/bash
# dev: does the engine clock policy matter for this latency-bound path?  clocks while the bench runs, then the bench under perf level "high"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2_clocks; rm -rf $O; mkdir -p $O
rocm-smi --showperflevel --showclocks > $O/idle.txt 2>&1
( timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_auto.json 2>/dev/null ) &
BP=$!
sleep 45; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" ; sleep 3; done > $O/during_auto.txt
wait $BP
rocm-smi --setperflevel high > $O/set_high.txt 2>&1
rocm-smi --showperflevel --showclocks >> $O/set_high.txt 2>&1
( timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_high.json 2>/dev/null ) &
BP=$!
sleep 45; for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" ; sleep 3; done > $O/during_high.txt
wait $BP
rocm-smi --setperflevel auto >> $O/set_high.txt 2>&1
for f in auto high; do python - <<PY
import json; d=json.load(open("$O/bench_$f.json")); print("$f", round(d["value"]), [round(k["avg_launch_us"],2) for k in d["roofline"]["kernels"][:4]])
PY
done
cat $O/idle.txt | grep -E "sclk|mclk|fclk|Perf" ; echo ---; cat $O/during_auto.txt | head -12; echo ---; grep -E "sclk|Perf|rror|denied" $O/set_high.txt | head; echo ---; head -8 $O/during_high.txt
