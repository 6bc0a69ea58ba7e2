import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 4, 2048
a, c = pair(1236, B, N, N)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
match = torch.empty(B, N, N, device=dev); temp = torch.empty(B, 4 * N, device=dev)
nbytes = L.pcc_approxmatch_workspace_bytes(B, N, N)
ws = torch.zeros(nbytes // 4 + 4, device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
rc = L.pcc_approxmatch_ws(B, N, N, t1.data_ptr(), t2.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, st)
torch.cuda.synchronize()
print('rc', rc)
w = ws.cpu().numpy()
n4 = N; nb = N // 16
def up(v): return (v + 15) & ~15
o = 0
soa1 = o; o = up(o + B * 3 * n4 * 4)
soa2 = o; o = up(o + B * 3 * n4 * 4)
rank1 = o; o = up(o + B * N * 4)
rank2 = o; o = up(o + B * N * 4)
box1 = o; o = up(o + B * nb * 8 * 4)
s = w[soa1 // 4: soa1 // 4 + 3 * n4].reshape(3, n4)
r = w[rank1 // 4: rank1 // 4 + N].view(np.int32)
bx = w[box1 // 4: box1 // 4 + nb * 8].reshape(nb, 8)
print('rank is a permutation:', np.array_equal(np.sort(r), np.arange(N)))
print('soa[rank[k]] == xyz[k]:', np.array_equal(s[:, r].T, a[0]))
ext = bx[:, 4:7] - bx[:, 0:3]
print('block box extent mean', ext.mean(0), 'max', ext.max(0), 'cloud extent', a[0].max(0) - a[0].min(0))
# tile boxes (128 pts)
tl = bx.reshape(nb // 8, 8, 8)
lo = tl[:, :, 0:3].min(1); hi = tl[:, :, 4:7].max(1)
print('tile box extent mean', (hi - lo).mean(0))
# skip fraction at level 0 for set1 tiles vs set1 blocks (proxy)
cut2 = 151.0 / 23637.0
cnt = 0; tot = 0
for t in range(lo.shape[0]):
    dx = np.maximum(np.maximum(lo[t][None] - bx[:, 4:7], bx[:, 0:3] - hi[t][None]), 0)
    lb = (dx ** 2).sum(1)
    cnt += (lb > cut2).sum(); tot += nb
print('skip fraction level0', cnt / tot)
