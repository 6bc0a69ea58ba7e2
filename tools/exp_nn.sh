#!/bin/bash
cd "$(dirname "$0")/.."
for v in 0 1 2; do
  PCC_NN_DEBUG=$v python3 - <<PY 2>&1 | grep -v amdgpu
import os, sys, torch
sys.path.insert(0, '.')
from tests.util import pair
from pointcloudcounterfactual_amd import backend
a, c = pair(1236, 32, 2048, 2048, 'recon')
t1, t2 = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
from pointcloudcounterfactual_amd import _lib
import ctypes
L = _lib.lib
for _ in range(3): backend.ChamferEMD(t1, t2, True, True)
torch.cuda.synchronize()
L.pcc_profile_enable(1)
for _ in range(10): backend.ChamferEMD(t1, t2, True, True)
torch.cuda.synchronize()
us = ctypes.c_double(); n = ctypes.c_int()
L.pcc_profile_read(b'nn_sorted_kernel', ctypes.byref(us), ctypes.byref(n))
print('PCC_NN_DEBUG=$v nn_sorted_kernel', round(us.value, 1), 'us x', n.value)
PY
done
