"""Print the last launch group of a rocprofv3 *_kernel_trace.csv as a timeline (µs from the first kernel's start)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.1f %8.1f  %7.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-40:]))
