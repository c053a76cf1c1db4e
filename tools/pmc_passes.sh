#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the GPU box; counters in their own passes, never with
# sys/hip/hsa tracing).  Usage: tools/pmc_passes.sh <outdir-under-gpurun_out>
# KWS_HIP_LIB=<variant .so> in the environment profiles another build of the same ABI (e.g. the round-2 tile kernel:
# tools/build_variant.sh tile -DKWS_X_MFCC_TILE_KERNEL), for before / after counters from one box.
set -o pipefail
OUT=/root/repo/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python /root/repo/bench.py --steps 5 --warmup 1 --spinup 4 --cpu-sample 0 --configs none --no-parity ${KWS_BENCH_EXTRA:-}"   # 10 launches of each kernel, 4096 clips each
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1 || exit 1; }
run sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq_b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))  # kernel -> counter -> dispatch -> sum over instances
for f in glob.glob(os.path.join(out, "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "dscnn" if "dscnn" in n else "refine" if "mfcc_refine" in n else "mfcc" if "mfcc" in n else None
        if k: agg[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fh:
    fh.write("# per launch of 4096 clips (sum over the counter's instances, mean over launches); bench.py --steps 5 --warmup 1 --spinup 4 --configs none\n")
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:6s} {c:28s} {sum(v.values())/len(v):16.1f}  (launches={len(v)})"
            print(line); fh.write(line + "\n")
PY
