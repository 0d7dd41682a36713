# HBM-side bytes per launch (ROUND=r04 by default; round 3's numbers: profiles/r03_pmc_traffic_summary.txt) (FETCH_SIZE x2 for coalesced streams on gfx950 — profiles/r01_hbm_traffic.md — and WRITE_SIZE, KiB,
# separate passes) of the fixed-focus kernels with the single-plane derived copy: blend_p3 at configs 2, 3, 4 (rank), 5 and blend_stdx
# (STD, RGBA views) at configs 3 and 5 — does the chain's second fetch reach HBM?
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "2 planar TEN_WM" "3 planar TEN_WM" "4 planar TEN_WM" "5 planar TEN_WM" "3 rgba STD" "5 rgba STD"; do
  set -- $spec
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic_${ROUND:-r04}/c$1_$3_$c -o t -- python3 tools/run_p3.py $1 $2 6 $3 > gpurun_out/pmc_traffic_${ROUND:-r04}_$1_$3_$c.log 2>&1 || echo "config $1 $3 $c failed"
  done
done
python3 - ${ROUND:-r04} <<'PY' | tee gpurun_out/${ROUND:-r04}_pmc_traffic_summary.txt
import csv, glob, sys
rnd = sys.argv[1]
px = {2: 1920 * 1080, 3: 1920 * 1080, 4: 3840 * 2160, 5: 3840 * 2160}
nv = {2: (64, 64), 3: (225, 45), 4: (64, 32), 5: (225, 64)}
for cfg, method, kern, out_b in ((2, "TEN_WM", "blend_p3", 3), (3, "TEN_WM", "blend_p3", 3), (4, "TEN_WM", "blend_p3", 3), (5, "TEN_WM", "blend_p3", 3), (3, "STD", "blend_stdx", 4), (5, "STD", "blend_stdx", 4)):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(f"gpurun_out/pmc_traffic_{rnd}/c{cfg}_{method}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        v = v[2:] if len(v) > 2 else v          # the first launches read a cold cache and (TEN_WM) precede the phase re-tuning
        vals[c] = sum(v) / max(len(v), 1)
    need_r, need_w = nv[cfg][0] * px[cfg] * 3, nv[cfg][1] * px[cfg] * out_b
    rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    print(f"config {cfg} {method:6s} {kern:10s}: read {rd/1e6:8.1f} MB (layout needs {need_r/1e6:8.1f}), written {wr/1e6:8.1f} MB (needs {need_w/1e6:8.1f}); "
          f"total {(rd+wr)/1e6:8.1f} vs {(need_r+need_w)/1e6:8.1f} MB = x{(rd+wr)/(need_r+need_w):.3f}   [FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB, WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB]")
PY
