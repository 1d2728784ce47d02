// smpl_amd/csrc/specialize.h -- per-robot kernel specialisation.  The kinematic structure of a compiled model
// (joint kinds and origins, which link carries which sphere tree, the checked pairs) is turned into compile-time
// constants (model_compile.cpp model_const_header) and kernels.hip is compiled against them with hiprtc when a
// planning space is created: the joint loop of the collision kernels becomes straight-line code.  Results are
// bit-identical to the generic kernels (same operations, same order).  Code objects are cached in the process and
// on disk ($SMPLX_CACHE_DIR, else $XDG_CACHE_HOME/smpl_amd, else $HOME/.cache/smpl_amd, else /tmp).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <tuple>
#include <utility>

#include "device_types.h"

namespace smplx {

enum KernelId {
    K_STATE_PREP = 0, K_EXPAND, K_PIPE_PREP, K_PIPE_SETUP, K_PIPE_CONFIGS, K_PIPE_FINISH, K_SMALL_BATCH, K_EDGE_VALID,
    K_STATE_VALID, K_HEURISTIC, K_SPHERE_POSITIONS, K_SEARCH, K_COUNT
};

// a kernel to launch: the per-robot build (hipFunction_t from the hiprtc module) when present, else the generic
// kernel linked into the library
struct KernelRef {
    hipFunction_t fn = nullptr;
    const void* generic = nullptr;
};

struct KernelSet {
    KernelRef k[K_COUNT];
    bool specialized = false;
};

// generic kernels only
void generic_kernels(KernelSet& ks);

// fills ks with the per-robot kernels for `model` on the current device; on failure returns false with the reason
// in `why` and leaves ks generic.  Thread-safe.
bool specialized_kernels(const SmplxModelDev& model, KernelSet& ks, std::string& why);

// Launch with the kernel's own parameter types: every argument is converted to the declared type first, then
// passed by address (hipModuleLaunchKernel / hipLaunchKernel take untyped argument arrays).
template <typename... P, typename... A, size_t... I>
inline hipError_t launch_impl(const KernelRef& k, dim3 grid, dim3 block, size_t lds, hipStream_t st, std::tuple<P...>& vals,
                              std::index_sequence<I...>)
{
    void* argv[sizeof...(P)] = {(void*)&std::get<I>(vals)...};
    if (k.fn) return hipModuleLaunchKernel(k.fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, st, argv, nullptr);
    return hipLaunchKernel(k.generic, grid, block, argv, lds, st);
}

template <typename... P, typename... A>
inline hipError_t launch(const KernelRef& k, void (*)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... a)
{
    static_assert(sizeof...(P) == sizeof...(A), "argument count does not match the kernel's prototype");
    std::tuple<P...> vals{static_cast<P>(std::forward<A>(a))...};
    return launch_impl<P...>(k, grid, block, lds, st, vals, std::index_sequence_for<P...>{});
}

}  // namespace smplx
