# The three flagged passes of the focus map (rows, columns, exact) timed separately: a measurement build launches them one by one.
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flagged
for v in auto factored_direct; do
LFI_AB_LIB=gpurun_ab/liblfi_meas.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/flagged/$v -o p -- python3 tools/run_focus.py $v 15 3840 2160 scene > gpurun_out/flagged/$v.log 2>&1 || echo failed
python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/flagged/{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last focus_map call: everything after the last focus_plan_shifts
last = max(i for i, r in enumerate(rows) if "focus_plan_shifts" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
print("==", sys.argv[1])
for r in rows[last:]:
    print("%-34s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-34:], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done
