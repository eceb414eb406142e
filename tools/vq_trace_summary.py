"""Median duration per VQ kernel and grid out of a rocprofv3 kernel trace (tools/gpu_vq_nearest_probe.py)."""
import csv, collections, re, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    m = re.search(r'(vq_\w+|p3_split_kernel|row_sqnorm_kernel)', n)
    if not m: continue
    key = (m.group(1) + ('<' + n.split('<')[1].split('>')[0] + '>' if 'x3_kernel<' in n else ''), r['Grid_Size_X'], r['Grid_Size_Y'])
    d.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    if v[len(v) // 2] > 40: print(k, len(v), 'med %.0f us' % v[len(v) // 2])
