# A/B of the focus map's range pass at 4K (15x15 scene): "auto" (focus_range_t where it applies) against "factored_direct" (focus_range).
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/range_ab
for v in auto factored_direct; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/range_ab/$v -o p -- python3 tools/run_focus.py $v 15 3840 2160 scene > gpurun_out/range_ab/$v.log 2>&1 || echo "$v failed"
  f=$(find gpurun_out/range_ab/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && python3 tools/kstats.py $f | head -12
done
