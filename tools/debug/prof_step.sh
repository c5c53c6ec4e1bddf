#!/bin/bash
# rocprofv3 kernel stats of a few training steps (GPU box): tools/debug/prof_step.sh <tag> [grep-pattern]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/time_bwd_kernels.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${2:-.}" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{r["Name"][:110]:110s} {r["Calls"]:>5s} {float(r["AverageNs"])/1e3:8.1f} us')
PY
