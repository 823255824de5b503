#!/bin/bash
# dev: is the sparse evaluation's term pass bound by the texture-addresser / L1 path of its compute unit?  Busy and stall counters of TA / TCP for the
# chained sparse launches (one counter set per pass, no trace domains, COGAPS_NO_GRAPH=1 as tools/pmc_pass.sh).
#   TAG=r6_ta bash tools/pmc_ta_probe.sh      (through gpurun)
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-ta}; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z0-9_a-z]+|TCP_[A-Z0-9_a-z]+|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VMEM_RD|SQ_ACTIVE_INST_VMEM|SQ_WAIT_INST_ANY|SQ_INST_CYCLES_VMEM_RD|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK_CONFLICT)\b" | sort -u > $O/avail.txt; wc -l $O/avail.txt
ARGS=${ARGS:---sparse --genes 50000 --samples 12500 --steps 4 --warmup 1}
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1)); rm -rf /tmp/ta_$i
  COGAPS_NO_GRAPH=1 timeout -k 5 900 rocprofv3 --pmc $SET --kernel-include-regex "${KREGEX:-chain_sparse_kernel}" --output-format csv -d /tmp/ta_$i -- python $R/bench.py --no-cpu $ARGS > $O/bench_$i.json 2> $O/err_$i.txt
  F=$(find /tmp/ta_$i -name '*counter_collection.csv' | head -1)
  echo "set $i: $SET rc=$? file=$F"
  [ -n "$F" ] && python3 - "$F" <<'P' | tee -a $O/ta_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()): print("   %-40s %16.1f per dispatch" % (c, v / len(n[k])))
P
done <<< "${SETS_TEXT:-GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY}"
