R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq2
mkdir -p $OUT/a
cd /tmp && export TMPDIR=/tmp
MVHDP_NARROW_WIDE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 $R/tools/per_sweep_times.py --sweeps 10 > $OUT/a.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sweep_fast_kernel<" in n:
        key = n.split("(")[0]
        agg.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print("   ", c, [round(x/1e6,1) for x in v])
kt = glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(kt)):
    if "sweep_fast_kernel<" in r["Kernel_Name"]:
        d.setdefault(r["Kernel_Name"].split("(")[0], []).append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2))
print(d)
PY
