// Launch wrappers of the Winograd layer kernel (kernels/conv_winograd_f32.h).  In the product build they are compiled in a translation
// unit of their own (aae_wino.hip, in parallel with the rest: the four phase bodies x two block geometries are a fifth of the library's
// compile time); the experiments build and the CPU emulator include this header into their single translation unit.
#pragma once

namespace aae_host {

#ifdef AAE_WINO_DECLARATIONS_ONLY
void wino_layer_launch(int geom, int wide, unsigned grid, hipStream_t stream, const aae::ConvWinoLayerArgs& p);
void wino_layer_multi_launch(int geom, unsigned grid, hipStream_t stream, const aae::ConvWinoMultiArgs& p);
void wino_set_attributes();
#else
#ifdef AAE_WINO_TU
#define AAE_WINO_LINKAGE
#else
#define AAE_WINO_LINKAGE static
#endif
AAE_WINO_LINKAGE void wino_layer_launch(int geom, int wide, unsigned grid, hipStream_t stream, const aae::ConvWinoLayerArgs& p) {
#ifdef AAE_EXPERIMENTS
    if (wide) {      // blocks of 4 waves over both 32-channel halves: measured 13 % slower than two waves per SIMD (tools/ubench/wino_layer_time.hip)
        if (geom == 0) AAE_LAUNCH((aae::conv_wino_layer_kernel<0, true>), dim3(grid), dim3(256), aae::wino_layer_smem_bytes<0>(), stream, p);
        else AAE_LAUNCH((aae::conv_wino_layer_kernel<1, true>), dim3(grid), dim3(256), aae::wino_layer_smem_bytes<1>(), stream, p);
        return;
    }
#endif
    (void)wide;
    if (geom == 0) AAE_LAUNCH((aae::conv_wino_layer_kernel<0, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<0>(), stream, p);
    else AAE_LAUNCH((aae::conv_wino_layer_kernel<1, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<1>(), stream, p);
}
AAE_WINO_LINKAGE void wino_layer_multi_launch(int geom, unsigned grid, hipStream_t stream, const aae::ConvWinoMultiArgs& p) {
    if (geom == 0) AAE_LAUNCH((aae::conv_wino_layer_multi_kernel<0>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<0>(), stream, p);
    else AAE_LAUNCH((aae::conv_wino_layer_multi_kernel<1>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<1>(), stream, p);
}
AAE_WINO_LINKAGE void wino_set_attributes() {
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_multi_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_multi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>());
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>());
#endif
}
#endif

}  // namespace aae_host
