cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/focus_split; mkdir -p $out
export LFI_AB_LIB=gpurun_ab/liblfi_meas.so
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > $out/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/focus_split/t/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if "focus" in r["Kernel_Name"]]
t0=min(int(r["Start_Timestamp"]) for r in rows)
last=[r for r in rows][-14:]
for r in last:
    print(f'{r["Kernel_Name"][:60]:60s} start {(int(r["Start_Timestamp"])-t0)/1e3:10.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  stream {r.get("Stream_Id","")}')
PY
