# kernel durations of the linearisation at both sizes
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python $GRAFT_REPO_ROOT/tools/lba_lin_sizes.py > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
d=defaultdict(list)
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if not ('k_lin' in k or 'k_reduce' in k): continue
    v2=sorted(v)
    big=[x for x in v if x > 2.5*v2[0]]
    small=[x for x in v if x <= 2.5*v2[0]]
    print(k, len(v), 'small median %.1f' % sorted(small)[len(small)//2], 'large median %.1f (%d)' % ((sorted(big)[len(big)//2] if big else 0), len(big)))
PY
