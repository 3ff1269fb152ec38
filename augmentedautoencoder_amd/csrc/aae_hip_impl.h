// Host side of libaae_hip.so: handle management, weight packing, launch planning.
// Compiled by hipcc for gfx950 through aae_hip.hip.  (tests/emu/ compiles the
// same text against a CPU fiber emulator to unit-test the launch logic; that
// build is test infrastructure and is never loaded by the package.)
//
// The including translation unit provides: the HIP runtime API, the wave
// primitives of device_intrinsics.h and
//   AAE_LAUNCH(kernel, grid, block, smem_bytes, stream, args...)
#pragma once

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aae_hip_tuning.h"
#include "kernels/tile_f32.h"
#include "kernels/multi_launch.h"
#include "kernels/conv_igemm_f32.h"
#include "kernels/conv_wavek_f32.h"
#include "kernels/conv_igemm_x3h.h"
#include "kernels/conv_winograd_f32.h"
#include "aae_wino_launch.h"
#include "kernels/conv_first_f32.h"
#include "kernels/conv_direct_generic.h"
#include "kernels/dense_gemv_f32.h"
#include "kernels/codebook_scan_f32.h"
#include "kernels/codebook_scan_bf16.h"
#include "kernels/codebook_scan_resident.h"
#include "kernels/crop_resize_u8.h"
#ifdef AAE_EXPERIMENTS
#include "kernels/detect_chain.h"
#endif
#include "aae_host_types.h"
#include "aae_encoder_plan.h"
#include "aae_encoder_launch.h"
#include "aae_codebook_scan.h"
#include "aae_abi_impl.h"

#include "aae_multi_impl.h"
#include "aae_decoder_impl.h"
