# SQ counters of the focus-map passes at 4K (15x15 scene): where does focus_range's time go?
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/r03_pmc_fr/p$i -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > gpurun_out/r03_pmc_fr_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03_pmc_fr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in agg:
    if "focus_range" in k or "focus_flagged" in k or "focus_pick" in k:
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-32s %16.0f" % (c, sum(v) / len(v)))
PY
