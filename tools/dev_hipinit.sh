#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1"; timeout 300 python -c "$2" 2>&1 | grep -v amdgpu.ids | tail -4; }
run "load, torch.is_available, current_device" "from cogaps_amd import _capi; L=_capi.load(); import torch; print(torch.cuda.is_available()); print(_capi.current_device())"
run "load, torch import only, current_device" "from cogaps_amd import _capi; L=_capi.load(); import torch; print(_capi.current_device())"
run "load, torch.is_available, session" "import numpy as np; from cogaps_amd import _capi; L=_capi.load(); import torch; print(torch.cuda.is_available()); S=_capi.Session(np.random.rand(50,20).astype('f4'), nPatterns=3, nIterations=10, seed=1); print('session ok')"
run "load, current_device, torch.is_available, torch tensor" "from cogaps_amd import _capi; L=_capi.load(); print(_capi.current_device()); import torch; print(torch.cuda.is_available()); print(torch.ones(3, device='cuda').sum().item())"
run "load, torch tensor, session" "import numpy as np; from cogaps_amd import _capi; L=_capi.load(); import torch; print(torch.ones(3, device='cuda').sum().item()); S=_capi.Session(np.random.rand(50,20).astype('f4'), nPatterns=3, nIterations=10, seed=1); print('session ok')"
run "maps" "from cogaps_amd import _capi; L=_capi.load(); import torch; torch.cuda.is_available(); print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l}))"
run "maps torch first" "import torch; torch.cuda.is_available(); from cogaps_amd import _capi; L=_capi.load(); print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l}))"
