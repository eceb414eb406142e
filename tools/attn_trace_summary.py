"""Median duration per attention kernel and probe shape out of a rocprofv3 kernel trace of tools/gpu_attn_probe.py."""
import csv, collections, re, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'attn' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = collections.OrderedDict()
for r in rows:
    n = re.search(r'(attn_\w+(<[\d, ]+>)?)', r['Kernel_Name']).group(1)
    d.setdefault((n, r['Grid_Size_X']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    runs = [v[i:i + 13] for i in range(0, len(v), 13)]
    print(k, ['%.0f' % sorted(x)[len(x) // 2] for x in runs])
