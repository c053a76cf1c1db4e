#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for l in ${LIBS:-base s1 s2}; do
  OUT=/root/repo/gpurun_out/pmc_mfcc_$l; mkdir -p $OUT
  KWS_HIP_LIB=/root/repo/tools/bin/libkws_$l.so timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p -- python /root/repo/tools/time_mfcc.py > $OUT/log.txt 2>&1
  python3 - $OUT $l <<'PY'
import csv,sys,glob,collections
out,l=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(list)
for f in glob.glob(out+"/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mfcc_i16" in r["Kernel_Name"]: agg[r["Counter_Name"]].append((r["Dispatch_Id"],float(r["Counter_Value"])))
for k,v in sorted(agg.items()):
    d=collections.defaultdict(float)
    for i,x in v: d[i]+=x
    print(l,k,sum(d.values())/len(d))
PY
done
