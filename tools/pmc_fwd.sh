#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS counters of the forward kernels, one rocprofv3 pass per group.
# usage: tools/pmc_fwd.sh <tag> [cfg] [batch] [kernel-substring]
set -u
TAG=${1:-dev}; CFG=${2:-B}; BATCH=${3:-1}; PAT=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcf_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_fwd.py $CFG $BATCH 6 > /dev/null 2> $OUT/p$i.log
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
        if not k.startswith("void vamp") and not k.startswith("vamp"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(acc):
        fh.write(k + "\n")
        for c in sorted(acc[k]):
            fh.write("    %-24s %16.0f  (avg over %d launches)\n" % (c, acc[k][c] / n[k][c], n[k][c]))
print(open(os.path.join(out, "summary.txt")).read())
PY
