# usage: r4_ab.sh "<variants>" "<workloads>"  — benches base + each variant lib on each workload, prints ms and kernel-group ms
VARS="$1"; WLS="${2:-c3 c3room}"
for wl in $WLS; do
  for v in base $VARS; do
    if [ $v = base ]; then unset BHRT_LIB; else export BHRT_LIB=$PWD/bhraytracer_amd/_variants/libbhrt_$v.so; fi
    python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-alone > gpurun_out/ab_${v}_${wl}.json 2>gpurun_out/ab_${v}_${wl}.err
    python - gpurun_out/ab_${v}_${wl}.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(j["ms_per_step"],2), {k:round(v/j["steps"]*1e3,2) for k,v in j["kernel_seconds"].items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  done
done
