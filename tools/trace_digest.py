"""rocprofv3 --stats csv -> the top-25 kernel table kept under profiles/."""
import csv, glob, sys
src, dst = sys.argv[1], sys.argv[2]
for f in glob.glob(src + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(dst, 'w') as out:
        for r in rows[:25]:
            line = "%-60s calls %6s  avg %10.1f ns  total %5.1f %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs']), float(r['Percentage']))
            print(line); out.write(line + "\n")
