# usage: r4_pmc.sh "<variants>" <workload>
VARS="$1"; WL=${2:-c3room}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  if [ $v = base ]; then unset BHRT_LIB; else export BHRT_LIB=$R/bhraytracer_amd/_variants/libbhrt_$v.so; fi
  rm -rf $R/gpurun_out/pmcA_$v $R/gpurun_out/pmcB_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmcA_$v -o p --output-format csv -- python3 $R/tools/pmc_workload.py $WL > $R/gpurun_out/pmcA_$v.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $R/gpurun_out/pmcB_$v -o p --output-format csv -- python3 $R/tools/pmc_workload.py $WL > $R/gpurun_out/pmcB_$v.log 2>&1
  python3 - $R/gpurun_out/pmcA_$v $R/gpurun_out/pmcB_$v $v <<'PY'
import csv,glob,sys,collections,os
for path in sys.argv[1:3]:
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(path,"**","*counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bhrt::","")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k in agg:
        if "stream" in k or "shadow_mesh" in k:
            print(sys.argv[3], k, {c:"%.4g"%v for c,v in sorted(agg[k].items())})
PY
done
