"""Timeline of one pcc_match_cost call from a rocprofv3 --kernel-trace csv: per kernel start / duration / gap to the
previous kernel on the same queue, and how much of the wall time has 0 / 1 / 2+ kernels running."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# the last complete call: from the last-but-one am_sort_kernel pair to the end
sorts = [i for i, r in enumerate(rows) if 'am_sort_kernel' in r['Kernel_Name']]
finishes = [i for i, r in enumerate(rows) if 'pair_finish_kernel' in r['Kernel_Name']]
lo = sorts[-2]; hi = finishes[-1]
call = rows[lo:hi + 1]
t0 = call[0]['s']
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.search(r'(am_\w+|pair_\w+|nn_\w+|chamfer_\w+)(<[^>(]*>)?', n)
    return m.group(0) if m else n[:40]
last_end = {}
print('start_us  dur_us  gap_us  queue  kernel')
for r in call:
    q = r.get('Queue_Id', '?')
    gap = (r['s'] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r['e']
    print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f} {gap:7.1f}  {q:>5}  {short(r['Kernel_Name'])}")
ev = sorted([(r['s'], 1) for r in call] + [(r['e'], -1) for r in call])
busy = {0: 0, 1: 0, 2: 0}
cur = 0; prev = ev[0][0]
for t, d in ev:
    busy[min(cur, 2)] += t - prev
    prev = t; cur += d
tot = (call[-1]['e'] - t0) / 1e3
print(f'call wall {tot:.1f} us; kernels running: 0 -> {busy[0] / 1e3:.1f} us, 1 -> {busy[1] / 1e3:.1f} us, 2+ -> {busy[2] / 1e3:.1f} us')
