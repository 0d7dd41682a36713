"""Timeline of the last lfi_focus_map call in a rocprofv3 kernel trace (start and duration per kernel, µs).  usage: focus_timeline.py <dir>"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "focus_plan_shifts" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s start %8.1f  end %8.1f  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-34:], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
