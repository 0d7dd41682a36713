# focus_line_keys without one of its parts ((R) rows, (C) columns, (A') pairs flagged on both axes): which one sets its 0.15 ms at 4K?
# Build first, here:  for m in 1 2 4; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DLK_SKIP=$m -o gpurun_ab/liblfi_lk$m.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl; done
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lk
for v in full 1 2 4; do
  lib=""; [ $v != full ] && lib=gpurun_ab/liblfi_lk$v.so
  LFI_AB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lk/$v -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > gpurun_out/lk/$v.log 2>&1 || echo "$v failed"
  f=$(find gpurun_out/lk/$v -name "*kernel_stats.csv" | head -1)
  echo "== without part $v"; [ -n "$f" ] && python3 tools/kstats.py $f 2>/dev/null | grep "focus_line_keys\|focus_filter\|focus_pick"
done
