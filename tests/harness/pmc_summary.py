"""Summarise the rocprofv3 outputs written by pmc.sh: per-kernel totals of every counter."""
import collections, csv, glob, json, sys
root = sys.argv[1]
out = {}
for f in glob.glob(root + '/kt/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        out.setdefault('kernel_stats', []).append({k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')})
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in glob.glob(root + '/*/*/*_counter_collection.csv'):
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
        meta[k] = {x: r[x] for x in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Workgroup_Size', 'Grid_Size')}
    for (k, c), d in disp.items():
        agg[k]['dispatches:' + c] = len(d)
out['counters'] = {k: dict(v, **meta[k]) for k, v in agg.items()}
print(json.dumps(out, indent=1))
