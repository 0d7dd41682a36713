# FETCH_SIZE / WRITE_SIZE of blend_p3 at BASELINE configs 3, 4 (rank) and 5 (planar layout): HBM bytes per launch against what the layouts need
# (gfx950: FETCH_SIZE x2 for coalesced streams, profiles/r01_hbm_traffic.md; KB units)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 3 4 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_tr/c${cfg}_$c -o t -- python3 tools/run_p3.py $cfg planar 4 > gpurun_out/pmc_tr_${cfg}_$c.log 2>&1 || echo "config $cfg $c failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
need = {3: (225 * 1920 * 1080 * 3, 45 * 1920 * 1080 * 3), 4: (64 * 3840 * 2160 * 3, 32 * 3840 * 2160 * 3), 5: (225 * 3840 * 2160 * 3, 64 * 3840 * 2160 * 3)}
for cfg in (3, 4, 5):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(f"gpurun_out/pmc_tr/c{cfg}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "blend_p3" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        v = v[1:] if len(v) > 1 else v
        vals[c] = sum(v) / max(len(v), 1)
    rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    print(f"config {cfg}: read {rd/1e6:8.1f} MB (layout needs {need[cfg][0]/1e6:8.1f}), written {wr/1e6:8.1f} MB (needs {need[cfg][1]/1e6:8.1f}); total {(rd+wr)/1e6:8.1f} vs {(sum(need[cfg]))/1e6:8.1f} MB = x{(rd+wr)/sum(need[cfg]):.3f}")
PY
