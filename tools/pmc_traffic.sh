# HBM bytes per launch of the default TEN_WM kernel (bench.py, config 2): FETCH_SIZE and WRITE_SIZE in separate passes (KiB;
# gfx950: FETCH_SIZE x2, calibrated in profiles/r01_hbm_traffic.md), plus the calibration copy from tools/ablate.
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic/$c -o t -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --prewarm-ms 0 > gpurun_out/traffic_$c.log 2>&1 || echo "$c failed"
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic/cal_$c -o t -- ./tools/ablate > gpurun_out/traffic_cal_$c.log 2>&1 || echo "cal $c failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/traffic/*")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if any(s in k for s in ("blend", "planar", "k_copy")):
            print("%-22s %-46s %-11s n=%3d mean %12.1f KiB = %8.1f MB" % (d.split("/")[-1], k, c, len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6))
PY
