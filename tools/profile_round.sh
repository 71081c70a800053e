#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 evidence of one round.  Usage: tools/profile_round.sh <tag> [workload]
# Writes under gpurun_out/<tag>/; tools/pmc_summary.py turns that into the files committed under profiles/.
# The counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -e
TAG=${1:-r01}; WL=${2:-c2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS="--steps 20 --warmup 3"; case $WL in c3|c4) STEPS="--steps 5 --warmup 1";; c3room|c5) STEPS="--steps 3 --warmup 1";; esac
# (1) the bench's timed region as it runs by default: in scenes with meshes the any-hit kernels of a wave step run BESIDE the next step's closest-hit kernels
rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL -o $WL --output-format csv -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-configs --no-alone $STEPS > $OUT/trace_$WL.log 2>&1
grep "^{\"metric\"" $OUT/trace_$WL.log | tail -1 > $OUT/bench_under_rocprof_$WL.json
# (2) every kernel with the GPU to itself (the counter passes serialise the kernels anyway): the durations the VALU issue fractions are formed with
export BHRT_SHADOW_OVERLAP=0
case $WL in c3|c3room|c4)
rocprofv3 --kernel-trace --stats -d $OUT/trace_alone_$WL -o $WL --output-format csv -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-configs --no-alone $STEPS > $OUT/trace_alone_$WL.log 2>&1
grep "^{\"metric\"" $OUT/trace_alone_$WL.log | tail -1 > $OUT/bench_under_rocprof_alone_$WL.json
;; esac
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$WL -o p --output-format csv -- python3 $R/tools/pmc_workload.py $WL > $OUT/pmc_fetch_$WL.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$WL -o p --output-format csv -- python3 $R/tools/pmc_workload.py $WL > $OUT/pmc_write_$WL.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d $OUT/pmc_sq_$WL -o p --output-format csv -- python3 $R/tools/pmc_workload.py $WL > $OUT/pmc_sq_$WL.log 2>&1
echo done $TAG $WL
