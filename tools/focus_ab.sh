# A/B of a debug environment switch of the factored focus estimate: per-kernel averages from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export LFI_FOCUS_STRIPES=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_focus_$v -o ff -- python3 tools/run_focus.py > gpurun_out/prof_focus_$v.log 2>&1
  echo "== LFI_FOCUS_STRIPES=$v"; python3 tools/kstats.py gpurun_out/prof_focus_$v/ff_kernel_stats.csv | grep -E "range|exact|pick|pad"
done
