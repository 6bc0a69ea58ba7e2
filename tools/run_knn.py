"""A few k-NN calls (c >= 4) for rocprofv3: argv = channels, 0/1/2 for the knn_nosplit switch, [batch]."""
import os, sys
os.environ.setdefault('PCC_TEST_HOOKS', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops, _lib
c, mode = int(sys.argv[1]), int(sys.argv[2])
b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
torch.manual_seed(0)
x = torch.randn(b, c, 2048, device='cuda:0')
_lib.set_tuning('knn_nosplit', mode)
for _ in range(5):
    idx = ops.hip_knn(x, 25)
torch.cuda.synchronize()
print(int(idx.sum()))
