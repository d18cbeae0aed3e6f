"""Summarise a rocprofv3 --pmc csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, d in acc.items():
            if 'cbfssm' not in k:
                continue
            print(k)
            for c, v in sorted(d.items()):
                print('   %-32s n=%3d mean=%.4g' % (c, len(v), sum(v) / len(v)))
