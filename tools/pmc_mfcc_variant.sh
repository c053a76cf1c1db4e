#!/bin/bash
# SQ / LDS counters of the MFCC kernel for the in-tree library and for A/B variants (GPU box; counters in their own pass,
# --kernel-trace only).   tools/pmc_mfcc_variant.sh <outdir-under-gpurun_out> default tools/bin/libkws_x.so ...
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  if [ "$lib" = default ]; then unset KWS_HIP_LIB; else export KWS_HIP_LIB=$REPO/$lib; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python $REPO/bench.py --model mfcc-only --steps 5 --warmup 1 --spinup 4 --cpu-sample 0 > "$OUT/$name.log" 2>&1 || { tail -3 "$OUT/$name.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob, os
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for d in sorted(glob.glob(os.path.join(out, "*/"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> sum over instances
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "kws_mfcc_i16_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for c, v in sorted(agg.items()):
            line = f"{os.path.basename(d.rstrip('/')):20s} {c:24s} {sum(v.values())/len(v):16.1f}  per launch (n={len(v)})"
            print(line); fh.write(line + "\n")
PY
