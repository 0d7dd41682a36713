# Round 4: where does the all-focus gather spend its time?  SQ / TA / TCP / TCC counters of the all-focus renders at config 5 (structured scene:
# estimated map and a constant map).  usage (GPU box): bash tools/pmc_allfocus_l1.sh [method=TEN_WM] [tag=r04]
: ${GRAFT_REPO_ROOT:?}
METHOD=${1:-TEN_WM}; TAG=${2:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  for which in estimated constant; do
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_pmc_afl1/${which}_p$i -o p -- python3 tools/run_allfocus.py $METHOD 3 $which > gpurun_out/${TAG}_pmc_afl1_${which}_$i.log 2>&1 || echo "pass $i $which failed"
  done
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for which in ("estimated", "constant"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}_pmc_afl1/{which}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        if "blend_" in k:
            print(which, k)
            for c, v in sorted(agg[k].items()):
                print("   %-40s %16.0f" % (c, sum(v) / len(v)))
PY
