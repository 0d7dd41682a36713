# Round 4: FETCH_SIZE / WRITE_SIZE of the all-focus STD render at config 5 (structured scene, estimated map) by blend_stdxa (chunks 2 and 3
# gathered twice) and by blend_afs (variant filtered_gather_once: every sample gathered once).
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for variant in auto filtered_gather_once; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r04_pmc_af/${variant}_$c -o t -- python3 tools/run_allfocus.py STD 3 estimated $variant > gpurun_out/r04_pmc_af_${variant}_$c.log 2>&1 || echo "$variant $c failed"
  done
done
python3 - <<'PY'
import csv, glob
for variant, kern in (("auto", "blend_stdxa"), ("filtered_gather_once", "blend_afs")):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(f"gpurun_out/r04_pmc_af/{variant}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        v = v[1:] if len(v) > 1 else v
        vals[c] = sum(v) / max(len(v), 1)
    need_r, need_w = 225 * 3840 * 2160 * 4, 64 * 3840 * 2160 * 4
    print(f"STD estimated {kern:12s}: FETCH_SIZE {vals['FETCH_SIZE']*1024/1e6:9.1f} MB (x2: {2*vals['FETCH_SIZE']*1024/1e6:9.1f}) against {need_r/1e6:.1f} MB of RGBA samples; WRITE_SIZE {vals['WRITE_SIZE']*1024/1e6:8.1f} MB against {need_w/1e6:.1f}")
PY
