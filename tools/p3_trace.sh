# Unit clocks of blend_p3 (tools/p3_trace.py) from a measurement build; run on the GPU box.  Results: gpurun_out/p3_trace.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_ab gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DLFI_P3_TRACE=1 ${P3_EXTRA} -shared -o gpurun_ab/liblfi_p3t.so lfinterpolator_amd/csrc/hip/lfi_hip.hip -ldl 2> gpurun_out/p3_trace_build.log || exit 1
LFI_AB_LIB=gpurun_ab/liblfi_p3t.so python3 tools/p3_trace.py $1 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/p3_trace${P3_TAG}.txt
