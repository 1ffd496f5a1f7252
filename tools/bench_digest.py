"""Print the handful of fields of a bench.py JSON line that the round's targets are stated in."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def find(o, k):
    if isinstance(o, dict):
        if k in o: return o[k]
        for v in o.values():
            r = find(v, k)
            if r is not None: return r
print("value", d['value'], "ms/step", d['ms_per_step'], "roofline", d['roofline']['frac'], "parity", d.get('parity'))
for k in ('config4_local_ba_optimize', 'tracking_per_frame_mean_of_scenes', 'pose_optimizer_2000_obs', 'config0_euroc_mono_init'):
    v = find(d, k)
    if isinstance(v, dict): v = {a: b for a, b in v.items() if a != 'note'}
    print(k, v)
