export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python /root/repo/tools/lba_lin_sizes.py > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
d=defaultdict(list)
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v2=sorted(v)
    print(k, len(v), 'median %.1f' % v2[len(v2)//2], 'p10 %.1f p90 %.1f' % (v2[len(v2)//10], v2[len(v2)*9//10]), 'last10 avg %.1f' % (sum(v[-10:])/10))
PY
