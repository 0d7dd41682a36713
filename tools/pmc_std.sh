# SQ counters of the STD production kernel (separate passes), averaged per launch; VARIANT=wave_m2_nt|persist_m2_nt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${VARIANT:-wave_m2_nt}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_std/p$i -o p -- python3 tools/run_variants.py $V STD > gpurun_out/pmc_std_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_std/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[r["Kernel_Name"].split("(")[0][-40:]]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in agg:
    if "blend" in k:
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-28s %14.0f" % (c, sum(v) / len(v)))
PY
