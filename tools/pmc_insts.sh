#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/insts; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 1 --warmup 0 --T 3 --no-profile > /dev/null 2> $OUT/err.txt
python - <<PY
import csv,re,glob,collections
f=glob.glob("$OUT/p/**/*counter_collection.csv",recursive=True)[0]
agg=collections.OrderedDict()
seen=set()
for r in csv.DictReader(open(f)):
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("irsde::",""); name=re.sub(r"\(.*","",name)[:70]
    a=agg.setdefault(name,collections.defaultdict(float))
    a[r["Counter_Name"]]+=float(r["Counter_Value"])
    k=(r["Dispatch_Id"])
    if (name,k) not in seen:
        seen.add((name,k)); a["n"]+=1; a["us"]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print("%-72s %6s %10s %12s %12s %8s %8s %8s"%("kernel","n","us","VALU(non-MFMA)","MFMA","VALU/MFMA","VMEM/MFMA","LDS/MFMA"))
for k,a in sorted(agg.items(), key=lambda kv:-kv[1]["us"])[:24]:
    m=a["SQ_INSTS_MFMA"]; v=a["SQ_INSTS_VALU"]-m
    print("%-72s %6d %10.0f %12.0f %12.0f %8.2f %8.2f %8.2f"%(k,a["n"],a["us"],v,m,v/m if m else 0,a["SQ_INSTS_VMEM"]/m if m else 0,a["SQ_INSTS_LDS"]/m if m else 0))
PY
rm -rf $OUT/p
