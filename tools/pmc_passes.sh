#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-kernel PMC counters of one training step, one rocprofv3
# pass per counter group (no trace domains), condensed to gpurun_out/pmc_<tag>.txt.
# usage: tools/pmc_passes.sh <tag> [cfg] [batch]
set -u
TAG=${1:-dev}; CFG=${2:-B}; BATCH=${3:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/time_bwd.py $CFG $BATCH > /dev/null 2> $OUT/p$i.log
done
cd $ROOT
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(acc):
        fh.write(k + "\n")
        for c in sorted(acc[k]):
            fh.write("    %-22s %14.0f  (avg over %d launches)\n" % (c, acc[k][c] / n[k][c], n[k][c]))
print(open(os.path.join(out, "summary.txt")).read())
PY
