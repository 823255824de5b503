#!/bin/bash
# dev: rebuild product + profile libs and the emulator builds, then run the quick emulator parity set.
set -e
R=/root/repo
make -C $R/cogaps_amd/csrc 2>&1 | grep -E "error" && exit 1
make -C $R/cogaps_amd/csrc libcogaps_hip_PROFILE_DEV.so 2>&1 | grep -E "error" && exit 1
for w in 64 256 1024; do make -C $R/tests/emul WIN=$w 2>&1 | grep -E "error" && exit 1; done
cd $R
python tools/dev_parity.py 256 modsim 300 | tail -1
python tools/dev_parity.py 64 gist 30 | tail -1
python tools/dev_parity_synth.py 256 5 6 2 200 | tail -1
python tools/dev_parity.py 1024 gist 12 | tail -1
python tools/dev_parity.py 256 gist 40 | tail -1
