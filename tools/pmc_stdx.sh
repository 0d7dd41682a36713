# SQ / LDS / memory counters of blend_stdx (config 5, fixed focus, STD, RGBA views): is the kernel paced by its vector instructions, the LDS, or
# by waiting?   usage (GPU box): bash tools/pmc_stdx.sh [TAG=r04] ; LFI_AB_LIB selects a measurement build.  Results: gpurun_out/TAG_pmc_stdx.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r04}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_pmc_stdx/p$i -o p -- python3 tools/run_p3.py 5 rgba 4 STD > gpurun_out/${TAG}_pmc_stdx_$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$TAG" <<'PY' | tee gpurun_out/$TAG\_pmc_stdx.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{sys.argv[1]}_pmc_stdx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    if "blend_stdx" in k:
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-32s %16.0f" % (c, sum(v[1:]) / max(len(v) - 1, 1)))
PY
