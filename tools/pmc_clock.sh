# effective shader clock during a kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_clk/std -o p -- python3 tools/run_variants.py persist_m2_nt STD > gpurun_out/pmc_clk_std.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_clk/ten -o p -- python3 tools/run_variants.py persist_m2_nt TEN_WM > gpurun_out/pmc_clk_ten.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_clk/*/*counter_collection.csv")):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "persist" in r["Kernel_Name"]:
            d = rows[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = float(r["Counter_Value"]); d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, d in rows.items():
        print(f.split("/")[2], "dispatch", k, "dur %.1f us" % (d["ns"] / 1e3), "GUI_ACTIVE/8 = %.0f cycles" % (d["GRBM_GUI_ACTIVE"] / 8),
              "clock %.2f GHz" % (d["GRBM_GUI_ACTIVE"] / 8 / d["ns"]), "MFMA busy/SIMD %.0f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024),
              "util %.2f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)))
PY
