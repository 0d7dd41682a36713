# FETCH_SIZE / WRITE_SIZE of the all-focus renders at config 5 (structured scene's estimated map, and a constant map): how much of the gather's
# traffic is sector over-fetch?  (FETCH_SIZE x2 on gfx950 for wide coalesced streams — NOT calibrated for 4-byte gathers: both factors are printed.)
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "TEN_WM estimated" "TEN_WM constant" "STD estimated"; do
  set -- $spec
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_af/$1_$2_$c -o t -- python3 tools/run_allfocus.py $1 4 $2 > gpurun_out/pmc_af_$1_$2_$c.log 2>&1 || echo "$spec $c failed"
  done
done
python3 - <<'PY'
import csv, glob
for method, which, kern in (("TEN_WM", "estimated", "blend_persist"), ("TEN_WM", "constant", "blend_persist"), ("STD", "estimated", "blend_stdxa")):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(f"gpurun_out/pmc_af/{method}_{which}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        v = v[1:] if len(v) > 1 else v
        vals[c] = sum(v) / max(len(v), 1)
    need_r, need_w = 225 * 3840 * 2160 * 4, 64 * 3840 * 2160 * 4
    print(f"{method:6s} {which:9s} {kern:13s}: FETCH_SIZE {vals['FETCH_SIZE']*1024/1e6:9.1f} MB (x2: {2*vals['FETCH_SIZE']*1024/1e6:9.1f}) against {need_r/1e6:.1f} MB of RGBA samples; WRITE_SIZE {vals['WRITE_SIZE']*1024/1e6:8.1f} MB against {need_w/1e6:.1f}")
PY
