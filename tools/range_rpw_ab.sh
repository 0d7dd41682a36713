# A/B of focus_range_t builds at 4K (15x15 scene): rows per reducing wave (FRT_RPW 4: eight reducing waves, tile 64 x 32; 2: twelve, tile 64 x 24)
# and units of LDS reads in flight per reducing wave (FRT_DEPTH 2..4).  Build the libraries first, here:
#   for v in "2 2" "2 3" "2 4"; do set -- $v; hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DFRT_RPW=$1 -DFRT_DEPTH=$2 \
#       -o gpurun_ab/liblfi_r$1d$2.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl; done
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rpw_ab
for v in ${VARIANTS:-product r2d2 r2d3 r2d4}; do
  lib=""; [ $v != product ] && lib=gpurun_ab/liblfi_$v.so
  LFI_AB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rpw_ab/$v -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > gpurun_out/rpw_ab/$v.log 2>&1 || echo "$v failed"
  f=$(find gpurun_out/rpw_ab/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && python3 tools/kstats.py $f | grep "focus_range\|focus_flagged\|focus_pick\|focus_line_keys\|focus_filter"
done
python3 tools/focus_timeline.py gpurun_out/rpw_ab/product
