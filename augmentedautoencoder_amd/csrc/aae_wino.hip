// Second translation unit of libaae_hip.so (product build): the polyphase-Winograd layer kernels (kernels/conv_winograd_f32.h) and their
// launch wrappers, compiled in parallel with aae_hip.hip -DAAE_SPLIT_WINO and linked with it (__graft_entry__.build()).
#include <hip/hip_runtime.h>

#include "device_intrinsics.h"

#define AAE_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)

#define AAE_WINO_TU
#include "kernels/conv_winograd_f32.h"
#include "aae_wino_launch.h"
