# SQ counters and effective clock of the non-tensor wavefront kernel (blend_std_vfma) at config 2
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_vfma/p$i -o p -- python3 tools/run_variants.py vfma STD > gpurun_out/pmc_vfma_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_vfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in agg:
    if "vfma" in k:
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
