#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ calibration for gathers: tools/ubench/gather.hip under separate rocprofv3 --pmc passes.
#   TAG=r6_gather bash tools/pmc_gather.sh      (through gpurun; the binary is built in the container: hipcc --offload-arch=gfx950 -O3 tools/ubench/gather.hip -o tools/ubench/gather_DEV.bin)
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-gather}; mkdir -p $O
$R/tools/ubench/gather_DEV.bin > $O/gather_plain.txt 2>&1; cat $O/gather_plain.txt
for C in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum"; do
  D=/tmp/pmcg_$(echo $C | tr ' ' '_'); rm -rf $D
  timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d $D -- $R/tools/ubench/gather_DEV.bin > $O/run_$(echo $C | tr ' ' '_').txt 2>&1
  echo "$C rc=$?"
done
python $R/tools/pmc_gather.py /tmp/pmcg_* > $O/fetch_size_calibration.txt 2>&1; cat $O/fetch_size_calibration.txt
