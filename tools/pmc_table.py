"""Per-kernel mean of every counter in rocprofv3 --pmc csv output: argv = output dirs ..., kernel-name substring."""
import csv, glob, sys, collections
dirs, pat = sys.argv[1:-1], sys.argv[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row['Kernel_Name']:
                acc[row['Kernel_Name'][:70]][row['Counter_Name']].append(float(row['Counter_Value']))
for kern, ctr in acc.items():
    print(kern)
    for name, v in sorted(ctr.items()):
        print(f'  {name:34s} {sum(v) / len(v):16.0f}   x{len(v)}')
