# Unit clocks of blend_stdx (tools/stdx_trace.py) from a measurement build; run on the GPU box.  Results: gpurun_out/stdx_trace.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_ab gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_SX_TRACE=1 ${SX_EXTRA} -shared -o gpurun_ab/liblfi_sxt.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/stdx_trace_build.log || exit 1
LFI_AB_LIB=gpurun_ab/liblfi_sxt.so python3 tools/stdx_trace.py 2>&1 | tee gpurun_out/stdx_trace${SX_TAG}.txt
