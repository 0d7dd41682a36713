# Where does blend_stdx's time go?  Measurement builds of the library (-DLFI_SX_ABL=n, see blend_stdx.hpp) timed on one box by
# tools/std15_time.py; run on the GPU box (hipcc is there too).  Results: gpurun_out/stdx_ablate.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_ab gpurun_out
for n in ${SX_ABL:-1 2 3 4 5}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_SX_ABL=$n -shared -o gpurun_ab/liblfi_sx$n.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/stdx_build_$n.log || exit 1
done
{ echo "== as built"; python3 tools/std15_time.py 2>&1 | grep stdx
  for n in ${SX_ABL:-1 2 3 4 5}; do echo "== LFI_SX_ABL=$n"; LFI_AB_LIB=gpurun_ab/liblfi_sx$n.so python3 tools/std15_time.py 2>&1 | grep stdx; done; } | tee gpurun_out/stdx_ablate.txt
