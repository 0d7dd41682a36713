# blend_stdx's RGBA stores: nontemporal vs ordinary (-DLFI_SX_NT=0), and every tile storing into the first tile's bytes (-DLFI_SX_ABL=6: the
# stores never leave the L2).  Run on the GPU box.  Results: gpurun_out/stdx_stores.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_ab gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_SX_NT=0 -shared -o gpurun_ab/liblfi_sxnt0.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/stdx_stores_build.log || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_SX_ABL=6 -shared -o gpurun_ab/liblfi_sx6.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2>> gpurun_out/stdx_stores_build.log || exit 1
{ echo "== as built"; python3 tools/std15_time.py 2>&1 | grep "views"
  echo "== ordinary stores"; LFI_AB_LIB=gpurun_ab/liblfi_sxnt0.so python3 tools/std15_time.py 2>&1 | grep stdx
  echo "== stores into one tile"; LFI_AB_LIB=gpurun_ab/liblfi_sx6.so python3 tools/std15_time.py 2>&1 | grep stdx; } | tee gpurun_out/stdx_stores.txt
