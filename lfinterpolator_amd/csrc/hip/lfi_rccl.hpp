// lfi_rccl.hpp — RCCL through dlopen: the library has no link-time dependency on it and single-GPU users never load it.
// Used by lfi_broadcast_grid (the one collective of the path: the light field broadcast once over xGMI, BASELINE.json north_star;
// the reference is single-GPU, src/interpolator.cu has no counterpart).  Included by lfi_hip.hip only.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

namespace {
struct Rccl
{
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

const Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if(!h)
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if(!h)
            return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Broadcast && x.GetErrorString;
        return x;
    }();
    return r;
}
} // namespace

