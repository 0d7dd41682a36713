# Where does blend_afs / blend_aft's time go?  Measurement builds of the library (-DLFI_AF_ABL=mask, see blend_af.hpp) timed on one box.
# usage (GPU box): bash tools/af_ablate.sh "1 2 4 8 3" [methods]     Results: gpurun_out/af_ablate.txt
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_ab gpurun_out
MASKS=${1:-"1 2 4 8"}; METHODS=${2:-STD}
for n in $MASKS; do
  [ -f gpurun_ab/liblfi_af$n.so ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_AF_ABL=$n -shared -o gpurun_ab/liblfi_af$n.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/af_build_$n.log || exit 1
done
{ echo "== as built"; python3 tools/af_time.py $METHODS 2>&1 | tail -1
  for n in $MASKS; do echo "== LFI_AF_ABL=$n"; LFI_AB_LIB=gpurun_ab/liblfi_af$n.so python3 tools/af_time.py $METHODS 2>&1 | tail -1; done; } | tee gpurun_out/af_ablate.txt
