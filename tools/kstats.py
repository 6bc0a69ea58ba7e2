#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    name = r['Name']
    m = re.search(r'(am_\w+|nn_\w+|reduce_\w+|knn_\w+|gather_\w+|scatter_\w+|sqnorm\w+|global_\w+|auction_\w+)(<[^>(]*>)?', name)
    print(f"{(m.group(0) if m else name[:60]):58s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
