#!/bin/bash
# Run ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence bench.py's numbers are judged
# against and writes it under gpurun_out/ (copy the summaries into profiles/ afterwards).
#   1. --kernel-trace --stats of the default bench command
#   2. PMC passes (FETCH_SIZE, then WRITE_SIZE; separate passes, no trace domains) of the same step
# usage: tools/collect_profiles.sh <round-tag> [cfg] [batch]
set -u
TAG=${1:-r01}; CFG=${2:-B}; BATCH=${3:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --cfg $CFG --batch $BATCH --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-matrix > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/time_bwd.py $CFG $BATCH > /dev/null 2> $OUT/pmc_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/time_bwd.py $CFG $BATCH > /dev/null 2> $OUT/pmc_write.log
cd $ROOT
python3 tools/summarize_profiles.py $OUT $TAG $CFG $BATCH
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write      # (the raw traces: gpurun merges at most 64 MiB back)
# the same three passes with early ray termination OFF (the data-independent path): VAMP_ERT=0 is read by
# vampire_amd.ops at import; exported here so that rocprofv3 launches python itself (no env wrapper)
export VAMP_ERT=0
OUT2=$ROOT/gpurun_out/profiles_${TAG}_noert
mkdir -p $OUT2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/stats -- python3 $ROOT/bench.py --cfg $CFG --batch $BATCH --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-matrix > $OUT2/bench_under_rocprof.json 2> $OUT2/stats.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT2/pmc_fetch -- python3 $ROOT/tools/time_bwd.py $CFG $BATCH > /dev/null 2> $OUT2/pmc_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT2/pmc_write -- python3 $ROOT/tools/time_bwd.py $CFG $BATCH > /dev/null 2> $OUT2/pmc_write.log
cd $ROOT
python3 tools/summarize_profiles.py $OUT2 ${TAG}_noert $CFG $BATCH > /dev/null
rm -rf $OUT2/stats $OUT2/pmc_fetch $OUT2/pmc_write
unset VAMP_ERT
