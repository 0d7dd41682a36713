# What a fixed-focus -f sweep costs the planar kernels (the derived copy's phases are stale: every render has new integer offsets): HBM bytes
# (FETCH_SIZE / WRITE_SIZE, separate passes), L1 tag accesses and L2 requests per launch, tuned against sweep, configs 2 and 5.
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
for cfg in 2 5; do for mode in tuned sweep; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_pmc_sweep/c${cfg}_${mode}_$i -o p -- python3 tools/run_sweep.py $cfg $mode planar TEN_WM 1 > gpurun_out/${TAG}_pmc_sweep_${cfg}_${mode}_$i.log 2>&1 || echo "$cfg $mode pass $i failed"
  done
done; done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
for cfg in (2, 5):
    for mode in ("tuned", "sweep"):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/{sys.argv[1]}_pmc_sweep/c{cfg}_{mode}_*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "blend_p3" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    agg["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print(f"config {cfg} {mode}: blend_p3, per launch (mean of {len(agg['FETCH_SIZE'])})")
        for c, v in sorted(agg.items()):
            m = sum(v) / len(v)
            extra = f"  = {m * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6:9.1f} MB{' (x2: gfx950 counts half of wide reads)' if c == 'FETCH_SIZE' else ''}" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
            print("   %-32s %16.0f%s" % (c, m, extra))
PY
