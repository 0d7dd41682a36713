# SQ counters of blend_p3 at a BASELINE config (separate passes), full kernel and the no-DMA ablation build; CONFIG=3 by default.
# Then FETCH_SIZE / WRITE_SIZE of config 2 in both view layouts (gfx950: FETCH_SIZE x2 for coalesced streams, profiles/r01_hbm_traffic.md).
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=${CONFIG:-3}
for abl in 0 2; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    LFI_P3_ABLATE=$abl timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_p3/c${C}_a${abl}_p$i -o p -- python3 tools/run_p3.py $C planar 5 > gpurun_out/pmc_p3_c${C}_a${abl}_$i.log 2>&1 || echo "config $C ablate $abl pass $i failed"
  done
done
for layout in planar rgba; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_p3/traffic_${layout}_$c -o t -- python3 tools/run_p3.py 2 $layout 5 > gpurun_out/pmc_p3_traffic_${layout}_$c.log 2>&1 || echo "$layout $c failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_p3/*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        if "blend" in k:
            print(d.split("/")[-1], k)
            for c, v in sorted(agg[k].items()):
                v = v[1:] if len(v) > 1 else v   # drop the first (cold) launch
                print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
