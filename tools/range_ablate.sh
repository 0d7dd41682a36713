# Where does focus_range_t's time go?  Measurement builds (-DFRT_ABL=1: no loads, =2: no reduction) against the full kernel, at 4K, ALONE
# (kernels serialised by the counter collection) — rocprofv3 kernel stats.
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/range_abl
for v in full abl1 abl2 abl3; do
  lib=""; [ $v != full ] && lib=gpurun_ab/liblfi_frt_$v.so
  LFI_AB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/range_abl/$v -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > gpurun_out/range_abl/$v.log 2>&1 || echo "$v failed"
  f=$(find gpurun_out/range_abl/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && python3 tools/kstats.py $f 2>/dev/null | grep "focus_range\|focus_flagged"
done
