import csv, glob, collections, sys
d=sys.argv[1]
f = glob.glob(f'{d}/*/*_counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    if 'sweep' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
tok = float(sys.argv[2])
for c,vals in sorted(agg.items()):
    m=sum(vals)/len(vals)
    print(f"   {c:24s} mean={m:.4g}  per-token={m/tok:.2f}")
f = glob.glob(f'{d}/*/*_kernel_trace.csv')[0]
durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f)) if 'sweep' in r['Kernel_Name']]
print('   durations ms', [round(x,2) for x in durs])
