# A/B of the focus-map kernels on ONE box: the built library against gpurun_ab/liblfi_hip_base.so (measurement build, see the commit
# that produced it): event-timed lfi_focus_map at config 5's shape, then rocprofv3 kernel stats of each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/focus_ab; mkdir -p $out
for round in 1 2; do
  python3 tools/focus_ab.py 20 >> $out/new.txt 2>&1 || exit 1
  LFI_AB_LIB=gpurun_ab/liblfi_hip_base.so python3 tools/focus_ab.py 20 >> $out/base.txt 2>&1 || exit 1
done
prof() { # name, env
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -o p -- python3 tools/run_focus.py auto 15 3840 2160 scene > $out/$1.log 2>&1 || echo "$1 failed"
  f=$(find $out/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/$1_kernel_stats.csv
}
prof new
export LFI_AB_LIB=gpurun_ab/liblfi_hip_base.so
prof base
unset LFI_AB_LIB
echo "--- new"; cat $out/new.txt; echo "--- base"; cat $out/base.txt
echo "--- new kernels"; head -12 $out/new_kernel_stats.csv; echo "--- base kernels"; head -12 $out/base_kernel_stats.csv
