// libaae_hip.so translation unit: gfx950 device code + the C ABI of include/aae_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC aae_hip.hip -o libaae_hip.so
// (__graft_entry__.build() compiles it with -DAAE_SPLIT_WINO next to aae_wino.hip -- the Winograd layer kernels in a translation unit of
//  their own, in parallel -- and links the two objects)
#include <hip/hip_runtime.h>

#include "device_intrinsics.h"

#define AAE_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)

// a launch whose blocks must all be resident at once (grid-wide waits inside the kernel): the host sizes its grid to the
// chip; on the GPU it is an ordinary launch (the CPU emulator of tests/emu/ runs such grids side by side)
#define AAE_LAUNCH_RESIDENT(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)

#if defined(AAE_SPLIT_WINO) && !defined(AAE_EXPERIMENTS)
#define AAE_WINO_DECLARATIONS_ONLY
#endif
#include "aae_hip_impl.h"
