#!/bin/bash
# Run ON THE GPU BOX: memory-pipeline counters of the forward kernels (each pass under `timeout`:
# some counter groups hang rocprofv3 on this pool).
# usage: tools/pmc_mem.sh <tag> [cfg] [kernel-substring] [group ...]
set -u
TAG=${1:-dev}; CFG=${2:-B}; PAT=${3:-}
shift 3 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_fwd.py $CFG 1 4 > /dev/null 2> $OUT/p$i.log || echo "group '$grp' failed/timeout"
done
cd $ROOT
python3 - "$OUT" "$PAT" <<'PY'
import collections, csv, glob, os, sys
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if pat and pat not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("    %-40s %16.0f  (avg over %d launches)" % (c, acc[k][c] / n[k][c], n[k][c]))
PY
