#!/bin/bash
# Evidence for profiles/ (run on the GPU box): the bench line, and for EVERY workload bench.py reports -- ds-cnn
# (headline), mfcc-only (configs[1]), cnn-trad-fpool3 (configs[2] literal), ds-cnn at 1024 clips (configs[3] shard),
# streaming push (configs[4]) -- rocprofv3 kernel statistics plus the two HBM-traffic counter passes (each counter in
# its own pass with --kernel-trace only, never combined with sys/hip/hsa tracing).
#   tools/collect_profiles.sh <round tag, e.g. r03> [workloads...]
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.  Fails if a kernel-stats file does not
# hold exactly spinup + warmup + steps calls of the workload's kernels.
set -o pipefail
TAG=${1:-r03}; shift
WORKLOADS=${*:-"ds-cnn mfcc-only cnn-trad-fpool3 ds-cnn-1024 stream"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$REPO" && timeout -k 10 500 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err" || { tail -5 "$OUT/bench_default.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
SPIN=60; WARM=20
for W in $WORKLOADS; do
  case $W in
    ds-cnn)          STEPS=200; CMD="python $REPO/bench.py --steps $STEPS --warmup $WARM --spinup $SPIN --cpu-sample 0 --configs none --no-parity"; KERNELS="kws_mfcc_i16_kernel kws_mfcc_refine_kernel kws_dscnn_fwd_kernel";;
    mfcc-only)       STEPS=200; CMD="python $REPO/bench.py --model mfcc-only --steps $STEPS --warmup $WARM --spinup $SPIN --cpu-sample 0"; KERNELS="kws_mfcc_i16_kernel kws_mfcc_refine_kernel";;
    cnn-trad-fpool3) STEPS=100; CMD="python $REPO/bench.py --model cnn-trad-fpool3 --steps $STEPS --warmup $WARM --spinup $SPIN --cpu-sample 0"; KERNELS="kws_mfcc_i16_kernel kws_cnntrad_conv_kernel kws_cnntrad_dense_kernel";;
    ds-cnn-1024)     STEPS=400; CMD="python $REPO/bench.py --batch 1024 --steps $STEPS --warmup $WARM --spinup $SPIN --cpu-sample 0 --configs none --no-parity"; KERNELS="kws_mfcc_i16_kernel kws_dscnn_fwd_kernel";;
    stream)          STEPS=300; CMD="python $REPO/tools/bench_stream.py 64 $STEPS eager"; KERNELS="kws_dscnn_fwd_kernel";;
    *) echo "unknown workload $W"; exit 1;;
  esac
  EXPECT=$((SPIN + WARM + STEPS)); [ "$W" = stream ] && EXPECT=$((STEPS + 60))   # bench_stream.py times the kernel over 60 more pushes
  echo "== $W: $CMD (expect $EXPECT calls per kernel)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/kt" -o kt -- $CMD > "$OUT/$W.kt.log" 2>&1 || { tail -5 "$OUT/$W.kt.log"; exit 1; }
  find "$OUT/$W/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_${W}_kernel_stats.csv" \;
  for K in $KERNELS; do
    CALLS=$(python3 - "$OUT/${TAG}_${W}_kernel_stats.csv" "$K" <<'PY'
import csv, sys
n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        n += int(r["Calls"])
print(n)
PY
)
    if [ "$CALLS" != "$EXPECT" ]; then echo "FAIL: $W: $K has $CALLS calls in the stats file, expected $EXPECT"; exit 1; fi
  done
  # counter passes: short runs (5 timed steps), one counter per pass
  case $W in
    stream) PCMD="python $REPO/tools/bench_stream.py 64 40 eager";;
    *)      PCMD="${CMD/--steps $STEPS --warmup $WARM --spinup $SPIN/--steps 5 --warmup 1 --spinup 4}";;
  esac
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$W/fetch" -o fetch -- $PCMD > "$OUT/$W.fetch.log" 2>&1 || { tail -5 "$OUT/$W.fetch.log"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$W/write" -o write -- $PCMD > "$OUT/$W.write.log" 2>&1 || { tail -5 "$OUT/$W.write.log"; exit 1; }
  find "$OUT/$W/fetch" -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_${W}_pmc_fetch_size.csv" \;
  find "$OUT/$W/write" -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_${W}_pmc_write_size.csv" \;
done
python3 - "$OUT" "$TAG" $WORKLOADS <<'PY'
import csv, json, sys, collections, os
out, tag, workloads = sys.argv[1], sys.argv[2], sys.argv[3:]
KEYS = ("kws_mfcc_i16_kernel", "kws_mfcc_refine_kernel", "kws_dscnn_fwd_kernel", "kws_cnntrad_conv_kernel", "kws_cnntrad_dense_kernel", "kws_stream_frame_kernel")
def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> dispatch -> sum over instances
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key in KEYS:
            if key in r["Kernel_Name"]:
                acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}
def stats(path):
    res = {}
    for r in csv.DictReader(open(path)):
        for key in KEYS:
            if key in r["Name"]:
                res[key] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return res
res = {"round": tag, "unit": "bytes per launch",
       "correction": "read = FETCH_SIZE KiB * 1024 * 2 (gfx950 half-count of wide reads); write = WRITE_SIZE KiB * 1024",
       "kernels": {}, "workloads": {}}
for w in workloads:
    fetch = per_kernel(f"{out}/{tag}_{w}_pmc_fetch_size.csv", "FETCH_SIZE")
    write = per_kernel(f"{out}/{tag}_{w}_pmc_write_size.csv", "WRITE_SIZE")
    st = stats(f"{out}/{tag}_{w}_kernel_stats.csv")
    ks = {}
    for k in fetch:
        rd, wr = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
        ks[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0), **st.get(k, {})}
    res["workloads"][w] = {"kernels": ks}
    if w == "ds-cnn":
        res["workload"] = "bench.py headline, 4096 clips per launch"
        res["kernels"] = ks
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
for w, d in res["workloads"].items():
    for k, v in d["kernels"].items():
        print(f"{w:16s} {k:26s} calls {v.get('calls')}  avg {v.get('avg_us', 0):9.2f} us  read {v['read_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:8.2f} MB")
PY
tail -c 600 "$OUT/${TAG}_bench_default.json"; echo
