# A/B of two builds of the library on one box, alternating processes: lfinterpolator_amd/lib/liblfi_hip.so (product) against LFI_AB_LIB (default
# lfinterpolator_amd/lib/liblfi_hip_base.so, built from another commit: git stash; hipcc … -o that path; git stash pop).  gpurun_out/p3_ab.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd "$GRAFT_REPO_ROOT" || exit 1
BASE=${1:-lfinterpolator_amd/lib/liblfi_hip_base.so}
{ for r in 1 2 3; do echo "== product"; python3 tools/p3_times.py 2>&1 | grep "config"
  echo "== base"; LFI_AB_LIB=$BASE python3 tools/p3_times.py 2>&1 | grep "config"; done; } | tee gpurun_out/p3_ab.txt
