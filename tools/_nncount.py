import os, sys, torch
sys.path.insert(0, '.')
from tests.util import pair
from pointcloudcounterfactual_amd import backend
a, c = pair(1236, 32, 2048, 2048, 'recon')
t1, t2 = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
out = backend.ChamferEMD(t1, t2, True, False, return_dist=True)
d1 = out[-2]
print('PCC_NN_DEBUG', os.environ.get('PCC_NN_DEBUG'), 'blocks scanned per group (beyond the seed): mean', float(d1.mean()), 'max', float(d1.max()), 'min', float(d1.min()))
