#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/<tag>_* into small tracked summaries under profiles/.

  profiles/<tag>_kernel_stats.csv   per-kernel calls / avg / total from --kernel-trace --stats
  profiles/<tag>_pmc.csv            per-kernel averages of the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*)
  profiles/pmc_summary.json         HBM bytes per launch for the dominant kernels, corrected as
                                    MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and under-reports
                                    wide coalesced reads by 2x on gfx950; WRITE_SIZE is exact)

  summarise_profiles.py <tag> [<out_tag> [nosummary]]: the secondary runs of tools/profile_bench.sh (<tag>x = the bench
  with its extra rows: k-NN, graph ops, pooling, auction; <tag>n = the no-skip probe) are written with
  `nosummary`, which leaves pmc_summary.json -- the headline run's -- alone.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
write_summary = not (len(sys.argv) > 3 and sys.argv[3] == 'nosummary')  # secondary runs keep pmc_summary.json (the headline's)
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)


def short(name: str) -> str:
    name = name.replace('(anonymous namespace)::', '').replace('void ', '', 1) if name.startswith('void ') else name.replace('(anonymous namespace)::', '')
    m = re.search(r'(am_\w+|nn_\w+|pair_\w+|chamfer_\w+|reduce_\w+|knn_\w+|gather_\w+|scatter_\w+|edge_\w+|sqnorm_\w+|global_\w+|graph_\w+|auction_\w+|emd_\w+|bn_\w+|nbrsum_\w+)(<[^>(]*>)?', name)
    if m:
        return m.group(0)
    return re.sub(r'\(.*', '', name)[:80] or 'unnamed'



def one(pattern: str) -> str | None:
    files = glob.glob(os.path.join(G, pattern))  # (a profiled command that starts children leaves one file per process)
    if not files:
        return None
    newest = max(os.path.getmtime(f) for f in files)  # (a directory may hold an earlier run of the same tag)
    recent = [f for f in files if newest - os.path.getmtime(f) < 120]
    return max(recent, key=os.path.getsize)


stats = one(f'{tag}_stats/*/*kernel_stats.csv')
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(P, f'{out_tag}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', 'percent'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                        f"{float(r['MaxNs']) / 1e3:.2f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}", r['Percentage']])
    print('wrote', f'profiles/{out_tag}_kernel_stats.csv')

pmc: dict[str, dict[str, list[float]]] = defaultdict(lambda: defaultdict(list))
for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_trans'):
    f = one(f'{tag}_{sub}/*/*counter_collection.csv')
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        pmc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
if pmc:
    counters = sorted({c for k in pmc.values() for c in k})
    with open(os.path.join(P, f'{out_tag}_pmc.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatches'] + [f'avg_{c}' for c in counters])
        for k, d in sorted(pmc.items()):
            n = max(len(v) for v in d.values())
            w.writerow([k, n] + [f'{sum(d[c]) / len(d[c]):.1f}' if d.get(c) else '' for c in counters])
    print('wrote', f'profiles/{out_tag}_pmc.csv')
    summary = {}
    for k, d in pmc.items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            key = re.sub(r'<.*', '', k)
            n = len(d['FETCH_SIZE'])  # dispatches of this instantiation: the family average is dispatch-weighted
            e = summary.setdefault(key, {'fetch_kib_raw': 0.0, 'write_kib': 0.0, 'valu_insts': 0.0, 'trans_insts': 0.0, 'dispatches': 0})
            e['fetch_kib_raw'] += sum(d['FETCH_SIZE'])
            e['write_kib'] += sum(d['WRITE_SIZE']) * n / max(len(d['WRITE_SIZE']), 1)
            if d.get('SQ_INSTS_VALU'):
                e['valu_insts'] += sum(d['SQ_INSTS_VALU']) * n / len(d['SQ_INSTS_VALU'])
            if d.get('SQ_INSTS_VALU_TRANS_F32'):
                e['trans_insts'] += sum(d['SQ_INSTS_VALU_TRANS_F32']) * n / len(d['SQ_INSTS_VALU_TRANS_F32'])
            e['dispatches'] += n
    # the 19 pass launches of one approxmatch are these two kernel families: one dispatch-weighted entry for both
    # (at this point the per-family fields still are sums over dispatches)
    fam = {'fetch_kib_raw': 0.0, 'write_kib': 0.0, 'valu_insts': 0.0, 'trans_insts': 0.0, 'dispatches': 0}
    for key in ('am_phase_kernel', 'am_fine_kernel'):
        for f in fam:
            fam[f] += summary.get(key, {}).get(f, 0)
    if fam['dispatches']:
        summary['am_phase+am_fine'] = fam
    for key, v in summary.items():
        n = v['dispatches']
        v['fetch_kib_raw'] /= n
        v['write_kib'] /= n
        # wave-level VALU instructions per launch (SQ_INSTS_VALU, summed over the chip): x64 = lane operations
        v['valu_insts_per_launch'] = v.pop('valu_insts') / n
        t = v.pop('trans_insts') / n
        v['trans_insts_per_launch'] = t if t > 0 else None  # v_exp_f32 / v_rsq_f32 / v_sqrt_f32 (4 issue slots each)
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (upper bound for
        # narrow accesses); WRITE_SIZE is exact.  Units are KiB.
        v['hbm_bytes_per_launch'] = (2.0 * v['fetch_kib_raw'] + v['write_kib']) * 1024.0
        v['source'] = f'profiles/{out_tag}_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes)'
    if write_summary:
        json.dump(summary, open(os.path.join(P, 'pmc_summary.json'), 'w'), indent=1)
        print('wrote profiles/pmc_summary.json')
