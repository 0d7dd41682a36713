# blend_stdx with the second workgroup of every CU started late (-DLFI_SX_PHASE=n: n × s_sleep 127 ≈ n × 8 k clocks): are the stalls of the
# chain units hidden when the two workgroups of a CU are out of phase?  Run on the GPU box.  Results: gpurun_out/stdx_phase.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_ab gpurun_out
for n in ${SX_PHASE:-1 2 17 18}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_SX_PHASE=$n -shared -o gpurun_ab/liblfi_sxp$n.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/stdx_phase_build_$n.log || exit 1
done
{ echo "== as built"; python3 tools/std15_time.py 2>&1 | grep stdx
  for n in ${SX_PHASE:-1 2 17 18}; do echo "== LFI_SX_PHASE=$n"; LFI_AB_LIB=gpurun_ab/liblfi_sxp$n.so python3 tools/std15_time.py 2>&1 | grep stdx; done; } | tee gpurun_out/stdx_phase.txt
