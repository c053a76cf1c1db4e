#!/bin/bash
# rocprofv3 counter passes for the cnn-trad-fpool3 kernels (GPU box).  Usage: tools/pmc_cnntrad.sh <outdir> [lib]
set -o pipefail
OUT=/root/repo/gpurun_out/${1:-pmc_ct}
[ -n "$2" ] && export KWS_HIP_LIB=/root/repo/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT" -o "$name" -- python /root/repo/tools/time_cnntrad.py > "$OUT/$name.log" 2>&1 || exit 1; }
run sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run sq_b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        math = "pair  " if "<true>" in n else "triple"   # kws_cnntrad_*_kernel<H2>
        k = (math + " conv") if "cnntrad_conv" in n else (math + " dense") if "cnntrad_dense" in n else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:13s} {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})"
            print(line); fh.write(line + "\n")
PY
