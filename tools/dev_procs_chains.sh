#!/bin/bash
# dev: P processes x C chains in flight on the one GPU; prints each process's aggregate (sum them)
P=$1; C=$2; IT=${3:-100}
cd /root/repo
for i in $(seq 1 $P); do ( timeout 280 python tools/dev_chains.py $C $IT 2>&1 | tail -1 ) & done
wait
