"""Print a rocprofv3 *_kernel_stats.csv as 'kernel  calls  average µs'."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"].split("(")[0][-48:].ljust(48), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3))
