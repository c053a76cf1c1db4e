#!/bin/bash
# Evidence for profiles/ (run on the GPU box): bench line, rocprofv3 kernel statistics, and the two HBM-traffic
# counter passes (each counter in its own pass, never combined with sys/hip/hsa tracing).
#   tools/collect_profiles.sh <round tag, e.g. r01>
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$REPO" && timeout -k 10 400 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python $REPO/bench.py --steps 200 --warmup 20 --cpu-sample 0 > "$OUT/kt.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/write.log" 2>&1 || exit 1
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
find "$OUT/fetch" -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_pmc_fetch_size.csv" \;
find "$OUT/write" -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_pmc_write_size.csv" \;
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> dispatch -> sum over instances
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for key in ("kws_mfcc_i16_kernel", "kws_dscnn_fwd_kernel"):
            if key in name:
                acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}
fetch = per_kernel(f"{out}/{tag}_pmc_fetch_size.csv", "FETCH_SIZE")
write = per_kernel(f"{out}/{tag}_pmc_write_size.csv", "WRITE_SIZE")
res = {"round": tag, "workload": "bench.py C3, 4096 clips per launch", "unit": "bytes per launch",
       "correction": "read = FETCH_SIZE KiB * 1024 * 2 (gfx950 half-count of wide reads); write = WRITE_SIZE KiB * 1024",
       "kernels": {}}
for k in fetch:
    rd, wr = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    res["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0)}
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
PY
grep -E "kws_|Name" "$OUT/${TAG}_bench_kernel_stats.csv" | head -8
tail -1 "$OUT/${TAG}_bench_default.json" | cut -c1-400
