"""A few pcc_match_cost calls (forward + gradients) at the bench size, for rocprofv3 runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048, sys.argv[1] if len(sys.argv) > 1 else 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
for _ in range(6):
    backend.MatchCostImplicit(t1, t2, True)
torch.cuda.synchronize()
