"""A few fused forward calls for a rocprofv3 --kernel-trace run (timeline analysis: tools/analyse_trace.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048, sys.argv[1] if len(sys.argv) > 1 else 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
for _ in range(8):
    backend.ChamferEMD(t1, t2, True, True)
torch.cuda.synchronize()
