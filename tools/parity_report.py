"""Measured parity of the approximate-EMD path against the oracle (GPU box): how far the HIP results sit from the
float64 recurrence, next to how far the oracle's own legitimate float32 evaluations (exp modes 0-3, see
oracle/structural_oracle.c) sit from it.  Prints one JSON line per case; the tolerances in
tests/test_gpu_structural.py are set from these numbers."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests.util import pair
from pointcloudcounterfactual_amd import backend

oracle.build(); oracle.set_threads(min(32, oracle.max_threads()))
dev = torch.device('cuda:0')
cases = [(2, 64, 64), (2, 257, 130), (2, 128, 256), (1, 513, 512), (2, 1024, 1024), (1, 2100, 2300), (4, 2048, 2048)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in s.split('x')) for s in sys.argv[1:]]
for (b, n, m) in cases:
    for kind in ('recon', 'uniform'):
        a, c = pair(300 + n + m, b, n, m, kind)
        t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
        match, _temp, cost = backend.ApproxMatchCost(t1, t2)
        ci, g1, g2 = backend.MatchCostImplicit(t1, t2, True)
        om64, _ = oracle.approxmatch_f64(a, c)
        oc64 = oracle.matchcost_f64(a, c, om64)
        h1, h2 = oracle.matchcostgrad_f64(a, c, om64)
        gscale = max(np.abs(h1).max(), np.abs(h2).max())
        got = match.cpu().numpy()
        rec = {'case': f'{b}x{n}x{m}', 'kind': kind,
               'elem_ours': float(np.abs(got - om64).max()),
               'cost_rel_ours': float(np.abs(cost.cpu().numpy() - oc64).max() / np.abs(oc64).max()),
               'cost_rel_implicit': float(np.abs(ci.cpu().numpy() - oc64).max() / np.abs(oc64).max()),
               'grad_ours': float(max(np.abs(g1.cpu().numpy() - h1).max(), np.abs(g2.cpu().numpy() - h2).max()) / gscale),
               'rowmass_ours': float(np.abs(got.sum(2) - om64.sum(2)).max()),
               'colmass_ours': float(np.abs(got.sum(1) - om64.sum(1)).max())}
        for mode in (0, 1, 2, 3):
            oracle.set_exp_mode(mode)
            om, _ = oracle.approxmatch(a, c)
            oc = oracle.matchcost(a, c, om)
            f1, f2 = oracle.matchcostgrad(a, c, om)
            rec[f'elem_m{mode}'] = float(np.abs(om - om64).max())
            rec[f'cost_rel_m{mode}'] = float(np.abs(oc - oc64).max() / np.abs(oc64).max())
            rec[f'grad_m{mode}'] = float(max(np.abs(f1 - h1).max(), np.abs(f2 - h2).max()) / gscale)
            rec[f'rowmass_m{mode}'] = float(np.abs(om.sum(2) - om64.sum(2)).max())
            rec[f'colmass_m{mode}'] = float(np.abs(om.sum(1) - om64.sum(1)).max())
        oracle.set_exp_mode(0)
        print(json.dumps(rec), flush=True)
