# Memory-side counters of the focus map's tail (focus_pick / focus_pick_sep, focus_line_keys) at 4K (15x15 scene): is the pick bound by HBM /
# fabric traffic, by the L2, by the L1's tags or by none of them?  One counter set per pass (FETCH_SIZE on gfx950: x2, MI355X_MICROARCH.md).
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05}; VARIANT=${2:-auto}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_pmc_tail_$VARIANT/p$i -o p -- python3 tools/run_focus.py $VARIANT 15 3840 2160 scene > gpurun_out/${TAG}_pmc_tail_${VARIANT}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$TAG" "$VARIANT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{sys.argv[1]}_pmc_tail_{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in agg:
    if "focus_pick" in k or "focus_line_keys" in k or "focus_range" in k:
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-32s %16.0f" % (c, sum(v) / len(v)))
PY
