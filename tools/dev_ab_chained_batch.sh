#!/bin/bash
# A/B of the batched multi-chain launches: one chained launch per step for all chains (chain_kernel_multi, round 6) against a generator launch + an evaluation launch per step (COGAPS_NO_CHAIN=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in ${CHAINS:-2 4 8 16}; do
  for rep in 1 2; do
    for mode in chained two_launches; do
      if [ $mode = two_launches ]; then E="COGAPS_NO_CHAIN=1"; else E="X=1"; fi
      env $E timeout 600 python bench.py --chains $C --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels']
print('chains %2d  %-12s %10.0f proposals/s  %6.1f ms/step   A: eval %5.1f gen %5.1f us   P: eval %5.1f gen %5.1f us' % ($C, '$mode', d['value'], d['ms_per_step'], k[0]['avg_launch_us'], k[1]['avg_launch_us'], k[2]['avg_launch_us'], k[3]['avg_launch_us']))"
    done
  done
done
