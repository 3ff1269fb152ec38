// The C ABI of include/aae_hip.h for the encoder and the codebook: argument checks, handle creation (weight upload / packing),
// option parsing, and the entry points that call forward_impl / run_scan.  Part of aae_hip_impl.h (the grouped multi-object
// query: aae_multi_impl.h; the decoder: aae_decoder_impl.h).
#pragma once

// =============================================================== C ABI
extern "C" {

int aae_abi_version(void) { return AAE_ABI_VERSION; }
const char* aae_last_error(void) { return aae_host::g_last_error.c_str(); }

int aae_encoder_create(const aae_encoder_desc* d, const void* const* hw, int n_weights, aae_encoder** out) {
    using namespace aae_host;
    if (!d || !hw || !out) return fail(AAE_ERR_INVALID, "aae_encoder_create: null argument");
    if (d->num_layers < 1 || d->num_layers > AAE_MAX_LAYERS)
        return fail(AAE_ERR_INVALID, "num_layers %d outside [1,%d]", d->num_layers, AAE_MAX_LAYERS);
    if (d->in_h < 1 || d->in_w < 1 || d->in_c < 1 || d->kernel_size < 1 || d->latent_size < 1)
        return fail(AAE_ERR_INVALID, "non-positive shape in encoder desc");
    const int per_layer = d->batch_norm ? 6 : 2;
    if (n_weights != d->num_layers * per_layer + 2)
        return fail(AAE_ERR_INVALID, "expected %d weight arrays, got %d", d->num_layers * per_layer + 2, n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!hw[i]) return fail(AAE_ERR_INVALID, "weight array %d is null", i);

    aae_encoder* enc = new aae_encoder();
    enc->desc = *d;
    auto bail = [&](int rc) { aae_encoder_destroy(enc); return rc; };

    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) enc->cu_count = cus;
    }
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float)((double)v / 255.0);   // float64 quotient, float32 feed cast
    if (int rc = upload(enc, lut, 256, &enc->lut)) return bail(rc);
    {
        const std::vector<float> zeros(kX3hRing + kX3hCaptured, 0.f);
        float* flags = nullptr;
        if (int rc = upload(enc, zeros.data(), zeros.size(), &flags)) return bail(rc);
        enc->x3h_sat = reinterpret_cast<int*>(flags);
    }

    int H = d->in_h, W = d->in_w, C = d->in_c, wi = 0;
    const float eps = d->bn_eps > 0.f ? d->bn_eps : 1e-3f;
    for (int li = 0; li < d->num_layers; ++li) {
        Layer L;
        L.index = li;
        L.H = H; L.W = W; L.Cin = C; L.Cout = d->num_filters[li]; L.KS = d->kernel_size; L.S = d->strides[li];
        if (L.Cout < 1 || L.S < 1) return bail(fail(AAE_ERR_INVALID, "layer %d: filters %d stride %d", li, L.Cout, L.S));
        same_pad(H, L.KS, L.S, &L.Ho, &L.pt);
        same_pad(W, L.KS, L.S, &L.Wo, &L.pl);
        L.CoutPad = (int)align_up((size_t)L.Cout, 128);
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(enc, k, (size_t)L.K() * L.Cout, &L.w_hwio)) return bail(rc);
        if (int rc = upload(enc, b, L.Cout, &L.bias)) return bail(rc);
        if (d->batch_norm) {
            const float* g = static_cast<const float*>(hw[wi++]);
            const float* be = static_cast<const float*>(hw[wi++]);
            const float* mu = static_cast<const float*>(hw[wi++]);
            const float* var = static_cast<const float*>(hw[wi++]);
            std::vector<float> sc(L.Cout), sh(L.Cout);
            for (int c = 0; c < L.Cout; ++c) {       // tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma
                const float inv = (1.0f / sqrtf(var[c] + eps)) * g[c];
                sc[c] = inv;
                sh[c] = be[c] - mu[c] * inv;
            }
            if (int rc = upload(enc, sc.data(), L.Cout, &L.bn_scale)) return bail(rc);
            if (int rc = upload(enc, sh.data(), L.Cout, &L.bn_shift)) return bail(rc);
        }
        if (li == 0 && first_layer_instantiated(L.KS, L.Cin)) {
            plan_first_layer(L);
            L.kind = (L.first_smem <= 160 * 1024 && L.first_packable) ? KIND_FIRST_MFMA : KIND_GENERIC;
        }
        if (L.kind == KIND_GENERIC && L.Cin % 32 == 0) {
            L.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, L.KS * L.KS, L.Cin, L.Cout, L.CoutPad);
            if (int rc = upload(enc, packed.data(), packed.size(), &L.wp)) return bail(rc);
            const std::vector<unsigned short> p16 = pack_weights_x3h(k, L.KS * L.KS, L.Cin, L.Cout, L.CoutPad, &L.w_shift);
            if (int rc = upload(enc, reinterpret_cast<const float*>(p16.data()), p16.size() / 2, reinterpret_cast<float**>(&L.wp16))) return bail(rc);
            // eligible for the polyphase-Winograd form?  Its weights (49/25 of the layer's weight bytes: 83.5 MB for the default network) are
            // prepared on first need -- ensure_winograd_weights, from the workspace-size queries -- so that a per-detection deployment
            // (30 T-LESS classes at 1 ... 16 boxes each never run the form) keeps 59 MB per object
            L.wino_geom = winograd_geometry(L);
        }
        enc->layers.push_back(L);
        H = L.Ho; W = L.Wo; C = L.Cout;
    }
    Layer& D = enc->dense;
    D.H = D.W = D.Ho = D.Wo = 1; D.KS = 1; D.S = 1; D.pt = D.pl = 0; D.relu = 0;
    D.Cin = H * W * C;                       // tf.layers.flatten, NHWC row-major
    D.Cout = d->latent_size;
    D.CoutPad = (int)align_up((size_t)D.Cout, 128);
    {
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(enc, b, D.Cout, &D.bias)) return bail(rc);
        if (D.Cin % 32 == 0) {
            D.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, 1, D.Cin, D.Cout, D.CoutPad);
            if (int rc = upload(enc, packed.data(), packed.size(), &D.wp)) return bail(rc);
            const std::vector<unsigned short> p16 = pack_weights_x3h(k, 1, D.Cin, D.Cout, D.CoutPad, &D.w_shift);
            if (int rc = upload(enc, reinterpret_cast<const float*>(p16.data()), p16.size() / 2, reinterpret_cast<float**>(&D.wp16))) return bail(rc);
        } else {
            D.kind = KIND_GENERIC;
            if (int rc = upload(enc, k, (size_t)D.K() * D.Cout, &D.w_hwio)) return bail(rc);
        }
    }
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#endif
    wino_set_attributes();
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 3, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 2, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 2, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<2, 2, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<0>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 3, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<1>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 2, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<1>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<3, 2, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<1>());
    (void)hipFuncSetAttribute((const void*)aae::conv_wino_phase_kernel<2, 2, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_smem_bytes<1>());
#endif
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#endif
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PARTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#endif
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
#ifdef AAE_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
#endif
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PARTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    if (enc->layers[0].kind == KIND_FIRST_MFMA) {
        const int sm = enc->layers[0].first_smem;
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    }
    *out = enc;
    return AAE_OK;
}

void aae_encoder_destroy(aae_encoder* enc) {
    if (!enc) return;
    for (void* p : enc->allocations) (void)hipFree(p);
    for (hipEvent_t e : enc->x3h_release_ev)
        if (e) (void)hipEventDestroy(e);
    delete enc;
}

int aae_encoder_set_option(aae_encoder* enc, const char* name, int value) {
    using namespace aae_host;
    if (!enc || !name) return fail(AAE_ERR_INVALID, "aae_encoder_set_option: null argument");
#ifndef AAE_EXPERIMENTS
    {
        // Kernel variants that measured slower than the defaults, and the profiling / ablation aids (one of which makes results wrong
        // on purpose), are compiled into the experiments build only (-DAAE_EXPERIMENTS: libaae_hip_experiments.so, tools/ and the A/B
        // tests).  Here their options accept the default value and nothing else.
        static const struct { const char* name; int only; } kExperimentOptions[] = {
            {"detect_chain", 0}, {"chain_timeline", 0}, {"wavek_timeline", 0}, {"wavek_ablate", 0}, {"wavek_waves", 4}, {"wavek_tiny_waves", 4},
            {"wavek_depth", 2}, {"wavek_pingpong", 0}, {"gemv_ticket", 1}, {"wavek_spread", 3}, {"igemm_dma", 1}, {"igemm_breg", 1}, {"x3h_dma", 1}, {"x3h_wide_min_blocks", 0}, {"winograd_wide", 0}};
        for (const auto& o : kExperimentOptions)
            if (!strcmp(name, o.name)) {
                if (value == o.only) return AAE_OK;
                return fail(AAE_ERR_UNSUPPORTED, "option '%s' = %d selects a kernel variant / profiling aid of the experiments build (-DAAE_EXPERIMENTS); this build runs '%s' = %d",
                            name, value, name, o.only);
            }
        if (!strcmp(name, "detect_chain_blocks")) return AAE_OK;
    }
#endif
    if (!strcmp(name, "splitk_min_base_blocks")) enc->splitk_min_base_blocks = value;
    else if (!strcmp(name, "splitk_target_blocks")) enc->splitk_target_blocks = value;
    else if (!strcmp(name, "reduce_small")) enc->reduce_small = value ? 1 : 0;
    else if (!strcmp(name, "igemm_stagger")) enc->igemm_stagger = value;
    else if (!strcmp(name, "x3h_dma")) enc->x3h_dma = value ? 1 : 0;
    else if (!strcmp(name, "x3h_wide256")) enc->x3h_wide256 = value ? 1 : 0;
    else if (!strcmp(name, "x3h_min_tiles")) enc->x3h_min_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "x3h_wide256_min_blocks")) enc->x3h_wide256_min_blocks = value < 1 ? 1 : value;
    else if (!strcmp(name, "x3h_wide_min_blocks")) enc->x3h_wide_min_blocks = value < 0 ? 0 : value;
    else if (!strcmp(name, "igemm_dma")) enc->igemm_dma = value ? 1 : 0;
    else if (!strcmp(name, "igemm_breg")) enc->igemm_breg = value ? 1 : 0;
    else if (!strcmp(name, "dense_gemv")) enc->dense_gemv = value ? 1 : 0;
    else if (!strcmp(name, "dense_gemv_max_batch")) enc->dense_gemv_max_batch = value;
    else if (!strcmp(name, "wavek_tail_split")) enc->wavek_tail_split = value ? 1 : 0;
    else if (!strcmp(name, "planner_cost_min_batch")) enc->planner_cost_min_batch = value < 1 ? 1 : value;
    else if (!strcmp(name, "planner_cost_batch3")) enc->planner_cost_batch3 = value ? 1 : 0;
    else if (!strcmp(name, "wavek_eff64x32_pct")) enc->wavek_eff64x32_pct = value < 30 ? 30 : (value > 100 ? 100 : value);
    else if (!strcmp(name, "wavek_g_boost")) enc->wavek_g_boost = value < 1 ? 1 : (value > 4 ? 4 : value);
    else if (!strcmp(name, "wavek_force_tail_tiles")) enc->wavek_force_tail_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_force_tail_g")) enc->wavek_force_tail_g = value < 2 ? 2 : value;
    else if (!strcmp(name, "gemv_ticket")) enc->gemv_ticket = value ? 1 : 0;
    else if (!strcmp(name, "wavek")) enc->wavek = value ? 1 : 0;
    else if (!strcmp(name, "wavek_dense")) enc->wavek_dense = value ? 1 : 0;
    else if (!strcmp(name, "wavek_ablate")) enc->wavek_ablate = value;
    else if (!strcmp(name, "wavek_balance")) enc->wavek_balance = value ? 1 : 0;
    else if (!strcmp(name, "planner_cost_model")) enc->planner_cost_model = value ? 1 : 0;
    else if (!strcmp(name, "ticket_prep")) enc->ticket_prep = value ? 1 : 0;
    else if (!strcmp(name, "multi_group_plan")) enc->multi_group_plan = value ? 1 : 0;
    else if (!strcmp(name, "multi_xcd_affine")) enc->multi_xcd_affine = value ? 1 : 0;
    else if (!strcmp(name, "multi_force_depth")) enc->multi_force_depth = value;
    else if (!strcmp(name, "multi_force_shape")) enc->multi_force_shape = value;
    else if (!strcmp(name, "multi_force_g")) enc->multi_force_g = value;
#ifdef AAE_EXPERIMENTS
    else if (!strcmp(name, "detect_chain")) {
        if (value) {
            // the persistent launch's grid barrier needs EVERY block resident: refuse the option unless the runtime confirms that one
            // 256-thread block with the chain's LDS footprint fits a compute unit and the device's CU count is known (a plain launch
            // of an over-sized grid would spin until its bounded wait traps).  A CU mask smaller than the device is not detectable
            // from here: the option stays opt-in.
            if (enc->cu_count <= 0) return fail(AAE_ERR_UNSUPPORTED, "detect_chain: the device's compute-unit count is unknown");
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)aae::detect_chain_kernel<1, 0, 0, 0>, 256, aae::kChainSmem);
            if (e != hipSuccess || per_cu < 1)
                return fail(AAE_ERR_UNSUPPORTED, "detect_chain: the runtime does not confirm residency of the persistent launch (%s, %d blocks per CU)",
                            e == hipSuccess ? "ok" : hipGetErrorString(e), per_cu);
        }
        enc->detect_chain = value ? 1 : 0;
    }
#endif
    else if (!strcmp(name, "detect_chain_blocks")) enc->detect_chain_blocks = value < 1 ? 1 : (value > aae_host::kChainMaxBlocks ? aae_host::kChainMaxBlocks : value);
    else if (!strcmp(name, "compact_workspace")) enc->compact_workspace = value ? 1 : 0;
    else if (!strcmp(name, "chain_timeline")) {
        if (value && !enc->wavek_timeline) {
            void* p = nullptr;
            AAE_HIP_TRY(hipMalloc(&p, 3 * 512 * 8 * sizeof(long long)));
            enc->allocations.push_back(p);
            enc->wavek_timeline = static_cast<long long*>(p);
        }
        enc->chain_timeline = value < 0 ? 0 : value;           // 1: phase edges of the launch; 1 + l: also the inner stamps of conv layer l (2 = conv2 ...)
        if (!value) enc->wavek_timeline = nullptr;
    }
    else if (!strcmp(name, "wavek_timeline")) {
        if (value && !enc->wavek_timeline) {
            void* p = nullptr;
            AAE_HIP_TRY(hipMalloc(&p, 3 * 512 * 8 * sizeof(long long)));
            enc->allocations.push_back(p);
            enc->wavek_timeline = static_cast<long long*>(p);
        }
        if (!value) enc->wavek_timeline = nullptr;     // (the buffer stays in `allocations` until the handle goes)
    }
    else if (!strcmp(name, "wavek_max_tiles")) enc->wavek_max_tiles = value < 0 ? 0 : (value > aae_host::kWaveKTileCap ? aae_host::kWaveKTileCap : value);
    else if (!strcmp(name, "wavek_narrow_max_tiles")) enc->wavek_narrow_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_target_blocks")) enc->wavek_target_blocks = value < 0 ? 0 : (value > 2 * aae_host::kLayerTicketWords ? 2 * aae_host::kLayerTicketWords : value);
    else if (!strcmp(name, "wavek_tiny_max_tiles")) enc->wavek_tiny_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_waves")) {
        if (value != 4 && value != 8) return fail(AAE_ERR_INVALID, "wavek_waves %d: 4 or 8", value);
        enc->wavek_waves = value;
    } else if (!strcmp(name, "wavek_pingpong")) enc->wavek_pingpong = value ? 1 : 0;
    else if (!strcmp(name, "wavek_spread")) enc->wavek_spread = value & 3;       // bit 0: 64 x 64 tiles, bit 1: 32 x 32 tiles (two accumulator chains)
    else if (!strcmp(name, "wavek_tiny_waves")) {
        if (value != 4 && value != 8) return fail(AAE_ERR_INVALID, "wavek_tiny_waves %d: 4 or 8", value);
        enc->wavek_tiny_waves = value;
    } else if (!strcmp(name, "wavek_depth")) {
        if (value != 2 && value != 3) return fail(AAE_ERR_INVALID, "wavek_depth %d: 2 or 3", value);
        enc->wavek_depth = value;
    }
    else if (!strcmp(name, "igemm_breg_wide")) enc->igemm_breg_wide = value ? 1 : 0;
    else if (!strcmp(name, "igemm_breg_wide_min_blocks")) enc->igemm_breg_wide_min_blocks = value;
    else if (!strcmp(name, "igemm_breg_min_blocks")) enc->igemm_breg_min_blocks = value;
    else if (!strcmp(name, "first_vec4")) enc->first_vec4 = value ? 1 : 0;
    else if (!strcmp(name, "first_group_split_max_tiles")) enc->first_group_split_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "first_target_blocks")) enc->first_target_blocks = value < 1 ? 1 : value;
    else if (!strcmp(name, "first_max_tiles_per_block")) enc->first_max_tiles_per_block = value < 1 ? 1 : value;
    else if (!strcmp(name, "x3h_act_shift")) {
        if (value < -8 || value > 12) return fail(AAE_ERR_INVALID, "x3h_act_shift %d outside [-8, 12]", value);
        enc->x3h_act_shift = value;
    } else if (!strcmp(name, "winograd")) {
        if (value != 0) {
            bool any = false;
            for (const Layer& L : enc->layers) any = any || L.wino_geom >= 0;
            if (!any) return fail(AAE_ERR_UNSUPPORTED, "winograd: no layer is eligible (5 x 5 stride-2 'SAME' layers behind the first one with 32 | Cin, 64 | Cout and 16 | Ho, Wo or an 8 x 8 output)");
        }
        if (value < 0 || value > 2) return fail(AAE_ERR_INVALID, "winograd %d: 0 = direct kernels, 1 = one launch per layer, 2 = one launch per polyphase component", value);
#ifndef AAE_EXPERIMENTS
        if (value == 2) return fail(AAE_ERR_UNSUPPORTED, "option 'winograd' = 2 (one launch per polyphase component: measured slower) is a kernel variant of the experiments build (-DAAE_EXPERIMENTS)");
#endif
        enc->winograd = value;
    } else if (!strcmp(name, "winograd_wide")) {
        enc->winograd_wide = value ? 1 : 0;
    } else if (!strcmp(name, "winograd_min_batch")) {
        enc->winograd_min_batch = value < 1 ? 1 : value;
    } else if (!strcmp(name, "winograd_min_fill_pct")) {
        enc->winograd_min_fill_pct = value < 1 ? 1 : (value > 100 ? 100 : value);
    } else if (!strcmp(name, "winograd_min_blocks")) {
        enc->winograd_min_blocks = value < 0 ? 0 : value;
    } else if (!strcmp(name, "multi_mid_group")) {
        enc->multi_mid_group = value ? 1 : 0;
    } else if (!strcmp(name, "multi_split_items")) {
        enc->multi_split_items = value ? 1 : 0;
    } else if (!strcmp(name, "multi_group_winograd")) {
        enc->multi_group_winograd = value ? 1 : 0;
    } else if (!strcmp(name, "multi_mid_scan")) {
        enc->multi_mid_scan = value ? 1 : 0;
    } else if (!strcmp(name, "multi_mid_ragged")) {
        enc->multi_mid_ragged = value ? 1 : 0;
    } else if (!strcmp(name, "winograd_xcd_cols")) {
        if (value < -1 || value > 8) return fail(AAE_ERR_INVALID, "winograd_xcd_cols %d: -1 = per-layer default, 0 = plain block order, 1 ... 8 = column blocks of a region per XCD", value);
        enc->winograd_xcd_cols = value;
    } else if (!strcmp(name, "precision")) {
        if (value < 0 || value > 2) return fail(AAE_ERR_INVALID, "precision %d: 0 = fp32, 1 = f32x3h, 2 = f32x3h where it is faster", value);
        if (value != 0) {
            bool ok = enc->layers[0].kind == KIND_FIRST_MFMA && enc->dense.kind == KIND_IGEMM;
            for (size_t i = 1; i < enc->layers.size(); ++i) ok = ok && enc->layers[i].kind == KIND_IGEMM;
            if (!ok) return fail(AAE_ERR_UNSUPPORTED, "f32x3h needs the matrix-core kernels on every layer (first layer 5x5 with C in {1,3}, later Cin %% 32 == 0)");
        }
        enc->precision = value;
    }
    else return fail(AAE_ERR_INVALID, "unknown encoder option '%s'", name);
    return AAE_OK;
}

size_t aae_encoder_workspace_bytes(const aae_encoder* enc, int B) {
    if (!enc || B < 1) return 0;
    // (a caller sizes its workspace for a batch before it runs it: the one place outside the hot calls that knows a batch is coming --
    //  batches that take the Winograd conv layers get their transformed weights here, once; a forward never allocates)
    if (aae_host::wants_winograd_weights(enc, B) && aae_host::ensure_winograd_weights(const_cast<aae_encoder*>(enc)) != AAE_OK) return 0;
    return aae_host::plan_workspace(enc, B).total;
}

int aae_encoder_forward(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                        size_t ws_bytes, void* stream) {
    aae_host::Timer tm;
    return aae_host::forward_impl(enc, x, x_dtype, B, z_out, workspace, ws_bytes, stream, tm);
}

int aae_encoder_forward_timed(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                              size_t ws_bytes, void* stream, float* kernel_ms, int max_kernels, int* n_kernels) {
    using namespace aae_host;
    if (!kernel_ms || !n_kernels) return fail(AAE_ERR_INVALID, "aae_encoder_forward_timed: null output");
    Timer tm;
    tm.on = true;
    int rc = forward_impl(enc, x, x_dtype, B, z_out, workspace, ws_bytes, stream, tm);
    if (rc == AAE_OK && !tm.ev.empty()) {
        hipError_t e = hipEventSynchronize(tm.ev.back());
        if (e != hipSuccess) rc = fail(AAE_ERR_RUNTIME, "hipEventSynchronize: %s", hipGetErrorString(e));
    }
    int n = (int)tm.ev.size() - 1;
    if (n < 0) n = 0;
    if (rc == AAE_OK) {
        for (int i = 0; i < n && i < max_kernels; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, tm.ev[i], tm.ev[i + 1]);
            kernel_ms[i] = ms;
        }
        *n_kernels = n;
    }
    for (hipEvent_t e : tm.ev) (void)hipEventDestroy(e);
    return rc;
}

const char* aae_encoder_kernel_label(const aae_encoder* enc, int i) {
    if (!enc || i < 0 || i >= (int)enc->records.size()) return "";
    return enc->records[i].label.c_str();
}

double aae_encoder_kernel_flops(const aae_encoder* enc, int i) {
    if (!enc || i < 0 || i >= (int)enc->records.size()) return 0.0;
    return enc->records[i].flops;
}

int aae_encoder_x3h_saturated(aae_encoder* enc, int* flag_out, void* stream_v) {
    using namespace aae_host;
    if (!enc || !flag_out) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_saturated: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    int v[kX3hRing + kX3hCaptured];
    AAE_HIP_TRY(hipMemcpyAsync(v, enc->x3h_sat, sizeof(v), hipMemcpyDeviceToHost, stream));
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    int any = 0;
    for (int i = 0; i < kX3hRing + kX3hCaptured; ++i) any |= v[i];
    if (any) AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat, 0, sizeof(v), stream));
    *flag_out = any ? 1 : 0;
    return AAE_OK;
}

int aae_encoder_x3h_last_slot(void) { return aae_host::t_x3h_last_slot; }

int aae_encoder_x3h_poll(aae_encoder* enc, const int* slots, int n, int* flags_out, void* stream_v) {
    using namespace aae_host;
    if (!enc || !slots || !flags_out || n < 0) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_poll: bad argument");
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= kX3hRing + kX3hCaptured) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_poll: slot %d out of range", slots[i]);
    if (n == 0) return AAE_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    int v[kX3hRing + kX3hCaptured];
    AAE_HIP_TRY(hipMemcpyAsync(v, enc->x3h_sat, sizeof(v), hipMemcpyDeviceToHost, stream));
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    for (int i = 0; i < n; ++i) {
        flags_out[i] = v[slots[i]] ? 1 : 0;
        if (v[slots[i]]) {                                   // (rare) clear it for the slot's next user
            AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat + slots[i], 0, sizeof(int), stream));
            v[slots[i]] = 0;
        }
    }
    return AAE_OK;
}

int aae_encoder_x3h_release_slot(aae_encoder* enc, int slot, void* stream_v) {
    using namespace aae_host;
    if (!enc) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: null handle");
    if (slot < kX3hRing || slot >= kX3hRing + kX3hCaptured) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: slot %d is not a captured forward's", slot);
    std::lock_guard<std::mutex> lk(enc->x3h_mu);
    if (slot >= kX3hRing + enc->x3h_captured || std::find(enc->x3h_free.begin(), enc->x3h_free.end(), slot) != enc->x3h_free.end())
        return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: slot %d is not in use", slot);
    // (the next owner starts from a lowered flag; asynchronous on the caller's stream: a synchronous null-stream memset is
    //  invalid while any stream of the process is being captured)
    AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat + slot, 0, sizeof(int), static_cast<hipStream_t>(stream_v)));
    // ... and the slot is handed out again only once that clear has executed (forward_impl asks the event): a capture on another stream that
    // took it earlier could replay before the clear and lose a raised flag to it
    const size_t k = (size_t)(slot - kX3hRing);
    if (enc->x3h_release_ev.size() <= k) enc->x3h_release_ev.resize(k + 1, nullptr);
    if (!enc->x3h_release_ev[k]) AAE_HIP_TRY(hipEventCreateWithFlags(&enc->x3h_release_ev[k], hipEventDisableTiming));
    AAE_HIP_TRY(hipEventRecord(enc->x3h_release_ev[k], static_cast<hipStream_t>(stream_v)));
    enc->x3h_free.push_back(slot);
    return AAE_OK;
}

int aae_encoder_debug_timeline(aae_encoder* enc, long long* host_out) {
    using namespace aae_host;
    if (!enc || !host_out) return fail(AAE_ERR_INVALID, "aae_encoder_debug_timeline: null argument");
    if (!enc->wavek_timeline) return fail(AAE_ERR_INVALID, "aae_encoder_debug_timeline: option wavek_timeline is off");
    AAE_HIP_TRY(hipDeviceSynchronize());
    AAE_HIP_TRY(hipMemcpy(host_out, enc->wavek_timeline, 3 * 512 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return AAE_OK;
}

int aae_has_experiments(void) {
#ifdef AAE_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int aae_encoder_split_precision_for_batch(const aae_encoder* enc, int B) { return (enc && B >= 1 && aae_host::runs_split(enc, B)) ? 1 : 0; }

int aae_encoder_activation_info(const aae_encoder* enc, int B, int layer, size_t* offset_bytes, size_t* count) {
    using namespace aae_host;
    if (!enc || !offset_bytes || !count || B < 1) return fail(AAE_ERR_INVALID, "aae_encoder_activation_info: bad argument");
    if (layer < 0 || layer >= (int)enc->layers.size()) return fail(AAE_ERR_INVALID, "layer %d out of range", layer);
    if (enc->compact_workspace && layer + 2 < (int)enc->layers.size())
        return fail(AAE_ERR_UNSUPPORTED, "compact_workspace: the output of layer %d has been overwritten by layer %d", layer, layer + 2);
    const Workspace ws = plan_workspace(enc, B);
    const Layer& L = enc->layers[layer];
    *offset_bytes = ws.act_off[layer];
    *count = (size_t)B * L.Ho * L.Wo * L.Cout;
    return AAE_OK;
}

int aae_codebook_create(const void* E, int N, int J, int dtype, int src_is_device, aae_codebook** out) {
    using namespace aae_host;
    if (!E || !out) return fail(AAE_ERR_INVALID, "aae_codebook_create: null argument");
    if (N < 1 || J < 1) return fail(AAE_ERR_INVALID, "codebook shape [%d,%d]", N, J);
    if (dtype != AAE_DTYPE_F32 && dtype != AAE_DTYPE_BF16)
        return fail(AAE_ERR_UNSUPPORTED, "codebook dtype %d: float32 (AAE_DTYPE_F32) and bfloat16 (AAE_DTYPE_BF16) are implemented", dtype);
    if (J % 4 != 0 || J > 128) return fail(AAE_ERR_UNSUPPORTED, "latent size %d: the scan kernels need J %% 4 == 0 and J <= 128", J);
    if (dtype == AAE_DTYPE_BF16 && J != 128) return fail(AAE_ERR_UNSUPPORTED, "latent size %d: the bf16 scan kernel is built for J == 128", J);
    if ((unsigned long long)N * J * sizeof(float) >= 0xFFFFFFF0ull) return fail(AAE_ERR_UNSUPPORTED, "codebook of %d x %d floats exceeds the 4 GiB buffer view", N, J);
    aae_codebook* cb = new aae_codebook();
    cb->N = N; cb->J = J; cb->dtype = dtype;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cb->cu_count = cus;
    }
    const size_t bytes = (size_t)N * J * (dtype == AAE_DTYPE_BF16 ? 2 : 4);
    // the rows start on a 2 MB boundary: a B <= 4 query streams the whole codebook in ~10 us and every block touches its 64 KB at once -- the
    // period moved by 10-20 % with where the allocation happened to land (profiles/r15/scan_identical_library_copies.jsonl)
    constexpr size_t kCodebookAlign = 2u << 20;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes + kCodebookAlign);
    if (e != hipSuccess) { delete cb; return fail(AAE_ERR_RUNTIME, "hipMalloc(codebook): %s", hipGetErrorString(e)); }
    cb->E_alloc = p;
    cb->E = reinterpret_cast<float*>(((uintptr_t)p + kCodebookAlign - 1) & ~(uintptr_t)(kCodebookAlign - 1));
    e = hipMemcpy(cb->E, E, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) { aae_codebook_destroy(cb); return fail(AAE_ERR_RUNTIME, "hipMemcpy(codebook): %s", hipGetErrorString(e)); }
    *out = cb;
    return AAE_OK;
}

static int gather_upright_rows(const aae_codebook* cb, const aae_codebook* sub, int stride, hipStream_t stream) {
    using namespace aae_host;
    aae::GatherRowsArgs g;
    g.src = cb->E; g.dst = sub->E; g.rows_out = sub->N; g.stride = stride;
    g.pieces_per_row = cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4) / 16;
    long long blocks = ((long long)g.rows_out * g.pieces_per_row + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    AAE_LAUNCH((aae::gather_rows_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, g);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_codebook_prepare_upright(aae_codebook* cb, int col_stride, void* stream_v) {
    using namespace aae_host;
    if (!cb) return fail(AAE_ERR_INVALID, "aae_codebook_prepare_upright: null handle");
    if (col_stride < 2) return fail(AAE_ERR_INVALID, "aae_codebook_prepare_upright: col_stride %d < 2", col_stride);
    if ((cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4)) % 16 != 0) return AAE_OK;     // rows are not 16-byte pieces: the masked scan stays in use
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (!cb->upright || cb->upright_stride != col_stride) {
        aae_codebook* sub = nullptr;
        for (auto& c : cb->upright_copies)
            if (c.first == col_stride) sub = c.second;
        if (!sub) {
            sub = new (std::nothrow) aae_codebook();
            if (!sub) return fail(AAE_ERR_RUNTIME, "out of host memory");
            sub->N = ceil_div(cb->N, col_stride); sub->J = cb->J; sub->dtype = cb->dtype;
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, (size_t)sub->N * sub->J * (sub->dtype == AAE_DTYPE_BF16 ? 2 : 4));
            if (e != hipSuccess) { delete sub; return fail(AAE_ERR_RUNTIME, "hipMalloc(upright codebook): %s", hipGetErrorString(e)); }
            sub->E = static_cast<float*>(p);
            cb->upright_copies.push_back({col_stride, sub});
        }
        sub->scan_mode = cb->scan_mode; sub->scan_ticket = cb->scan_ticket; sub->topk_prune = cb->topk_prune; sub->cu_count = cb->cu_count;
        sub->scan_walk = cb->scan_walk; sub->scan_fused_norm = cb->scan_fused_norm; sub->scan_rh4 = cb->scan_rh4; sub->scan_resident_fin = cb->scan_resident_fin;
        cb->upright = sub; cb->upright_stride = col_stride;
    }
    if (int rc = gather_upright_rows(cb, cb->upright, cb->upright_stride, stream)) return rc;
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    return AAE_OK;
}

int aae_codebook_update(aae_codebook* cb, const void* E, int src_is_device, void* stream) {
    using namespace aae_host;
    if (!cb || !E) return fail(AAE_ERR_INVALID, "aae_codebook_update: null argument");
    AAE_HIP_TRY(hipMemcpyAsync(cb->E, E, (size_t)cb->N * cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4),
                               src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    for (auto& c : cb->upright_copies)                                                          // keep every compacted copy in step
        if (int rc = gather_upright_rows(cb, c.second, c.first, static_cast<hipStream_t>(stream))) return rc;
    AAE_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return AAE_OK;
}

void aae_codebook_destroy(aae_codebook* cb) {
    if (!cb) return;
    for (auto& c : cb->upright_copies) aae_codebook_destroy(c.second);
    if (cb->E_alloc) (void)hipFree(cb->E_alloc);
    else if (cb->E) (void)hipFree(cb->E);
    delete cb;
}

int aae_codebook_set_scan_mode(aae_codebook* cb, int mode) {
    using namespace aae_host;
    if (!cb) return fail(AAE_ERR_INVALID, "aae_codebook_set_scan_mode: null handle");
    if (mode != AAE_SCAN_AUTO && mode != AAE_SCAN_GEMV && mode != AAE_SCAN_MFMA && mode != AAE_SCAN_STREAM && mode != AAE_SCAN_STREAM_2L &&
        mode != AAE_SCAN_AUTO_NO_PRUNE && mode != AAE_SCAN_STREAM_WALK && mode != AAE_SCAN_AUTO_PACKED && mode != AAE_SCAN_AUTO_RH2 && mode != AAE_SCAN_AUTO_FIN)
        return fail(AAE_ERR_INVALID, "scan mode %d", mode);
#ifndef AAE_EXPERIMENTS
    if (mode == AAE_SCAN_GEMV || mode == AAE_SCAN_STREAM_WALK)
        return fail(AAE_ERR_UNSUPPORTED, "scan mode %d (the round-1 shuffle-reduction scan / the walking stream scan: measured slower) exists in the experiments build only (-DAAE_EXPERIMENTS)", mode);
#endif
    cb->scan_ticket = mode == AAE_SCAN_STREAM_2L ? 0 : 1;
    cb->topk_prune = mode == AAE_SCAN_AUTO_NO_PRUNE ? 0 : 1;
    cb->scan_walk = mode == AAE_SCAN_STREAM_WALK ? 1 : 0;
    cb->scan_fused_norm = mode == AAE_SCAN_AUTO_PACKED ? 0 : 1;
    cb->scan_rh4 = mode == AAE_SCAN_AUTO_RH2 ? 0 : 1;
    cb->scan_resident_fin = mode == AAE_SCAN_AUTO_FIN ? 1 : 0;
    cb->scan_mode = (mode == AAE_SCAN_STREAM_2L || mode == AAE_SCAN_STREAM_WALK) ? AAE_SCAN_STREAM
                    : ((mode == AAE_SCAN_AUTO_NO_PRUNE || mode == AAE_SCAN_AUTO_PACKED || mode == AAE_SCAN_AUTO_RH2 || mode == AAE_SCAN_AUTO_FIN) ? AAE_SCAN_AUTO : mode);
    for (auto& c : cb->upright_copies) {
        c.second->scan_mode = cb->scan_mode; c.second->scan_ticket = cb->scan_ticket; c.second->topk_prune = cb->topk_prune; c.second->scan_walk = cb->scan_walk;
        c.second->scan_fused_norm = cb->scan_fused_norm; c.second->scan_rh4 = cb->scan_rh4; c.second->scan_resident_fin = cb->scan_resident_fin;
    }
    return AAE_OK;
}

size_t aae_codebook_workspace_bytes(const aae_codebook* cb, int B, int topk) {
    if (!cb || B < 1 || topk < 1) return 0;
    return std::max(aae_host::plan_scan(cb, B, topk).total, aae_host::plan_scan(cb, B, topk, true).total);    // (with or without a masked upright query)
}

// prepared_nonce != 0: the scan's ticket words (front of `workspace`) carry this nonce already
static int nn_impl(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                   float* score_out, void* workspace, size_t ws_bytes, void* stream_v, unsigned prepared_nonce) {
    using namespace aae_host;
    if (!cb || !z || !idx_out || !score_out) return fail(AAE_ERR_INVALID, "aae_codebook_nn: null argument");
    if (B < 1 || topk < 1 || topk > cb->N) return fail(AAE_ERR_INVALID, "aae_codebook_nn: B=%d topk=%d N=%d", B, topk, cb->N);
    if (col_stride < 1) return fail(AAE_ERR_INVALID, "col_stride %d < 1", col_stride);
    if (topk > 1 && col_stride != 1) return fail(AAE_ERR_INVALID, "upright (col_stride>1) is defined for topk == 1 only (codebook.py:65-66)");
    if (topk > 1 && B > 65535) return fail(AAE_ERR_UNSUPPORTED, "top-k for more than 65535 queries per call (got %d): split the batch", B);
    {
        const size_t need = std::max(plan_scan(cb, B, topk).total, plan_scan(cb, B, topk, true).total);
        if (ws_bytes < need) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    }
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    // upright: scan the prepared every-col_stride-th-row copy (1/col_stride of the work) and scale the row id back
    int idx_scale = 1;
    if (col_stride > 1 && cb->upright && cb->upright_stride == col_stride) {
        idx_scale = col_stride;
        cb = cb->upright;
        col_stride = 1;
    }
    const ScanPlan s = plan_scan(cb, B, topk, col_stride > 1);          // never larger than the plan of the full codebook
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    float* cs = (topk > 1 && !s.topk_fused) ? reinterpret_cast<float*>(base + s.cs_off) : nullptr;
    int partial_rows = s.nblk;
    // B <= 4, top-1 on a stream kernel: the last block to arrive merges the block partials -- the query is one launch
    ScanTicketOut fin;
    fin.idx_out = idx_out; fin.score_out = score_out; fin.idx_scale = idx_scale; fin.nonce = prepared_nonce;
    // (opt-in, AAE_SCAN_AUTO_FIN: the same for the query-resident scan of at most 32 queries -- one row of row blocks)
    const bool resident_fin = topk == 1 && cb->scan_resident_fin && s.resident_ok && ceil_div(s.Bpad, 256 / s.res_rh) == 1 && col_stride == 1;
    const bool one_launch = topk == 1 && ((s.stream && cb->scan_ticket != 0) || resident_fin);
    if (int rc = run_scan(cb, z, B, col_stride, cs, s, base, stream, &partial_rows, one_launch ? &fin : nullptr, s.topk_fused ? topk : 1)) return rc;
    if (one_launch) return AAE_OK;
    if (topk == 1) {
        aae::ArgmaxReduceArgs r;
        r.pval = reinterpret_cast<float*>(base + s.pval_off);
        r.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        r.idx_out = reinterpret_cast<long long*>(idx_out);
        r.score_out = score_out; r.nblk = partial_rows; r.B = B; r.Bstride = s.Bstride; r.idx_scale = idx_scale;
        AAE_LAUNCH((aae::argmax_reduce_kernel), dim3(B), dim3(256), 64, stream, r);
    } else {
        aae::TopKArgs t;
        t.cs = cs; t.idx_out = reinterpret_cast<long long*>(idx_out); t.score_out = score_out; t.N = cb->N; t.k = topk;
        t.chunks = s.cand_chunks;
        t.cand_v = reinterpret_cast<float*>(base + s.cand_off);
        t.cand_i = reinterpret_cast<int*>(base + s.cand_off + align_up((size_t)B * t.chunks * topk * sizeof(float), 256));
        if (!s.topk_fused) AAE_LAUNCH((aae::topk_chunks_kernel), dim3(t.chunks, B), dim3(256), 64, stream, t);   // (fused: the scan wrote the lists)
        if (t.chunks * topk <= 256 * aae::kTopKMergeSlots) AAE_LAUNCH((aae::topk_merge_kernel<true>), dim3(B), dim3(256), 64, stream, t);
        else AAE_LAUNCH((aae::topk_merge_kernel<false>), dim3(B), dim3(256), 64, stream, t);
    }
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_codebook_nn(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                    float* score_out, void* workspace, size_t ws_bytes, void* stream) {
    return nn_impl(cb, z, B, topk, col_stride, idx_out, score_out, workspace, ws_bytes, stream, 0u);
}

int aae_codebook_nn_timed(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                          float* score_out, void* workspace, size_t ws_bytes, void* stream_v, int reps, float* kernel_ms) {
    using namespace aae_host;
    if (!kernel_ms || reps < 1) return fail(AAE_ERR_INVALID, "aae_codebook_nn_timed: null output or reps < 1");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    hipEvent_t e0, e1;
    AAE_HIP_TRY(hipEventCreate(&e0));
    AAE_HIP_TRY(hipEventCreate(&e1));
    int rc = AAE_OK;
    hipError_t e = hipEventRecord(e0, stream);
    if (e == hipSuccess) {
        for (int r = 0; r < reps && rc == AAE_OK; ++r) rc = nn_impl(cb, z, B, topk, col_stride, idx_out, score_out, workspace, ws_bytes, stream_v, 0u);
        e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess && rc == AAE_OK) e = hipEventElapsedTime(kernel_ms, e0, e1);
        if (e == hipSuccess && rc == AAE_OK) *kernel_ms /= (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == AAE_OK && e != hipSuccess) rc = fail(AAE_ERR_RUNTIME, "aae_codebook_nn_timed: %s", hipGetErrorString(e));
    return rc;
}

int aae_encode_nn(aae_encoder* enc, aae_codebook* cb, const void* x, int x_dtype, int B, int col_stride, float* z_out,
                  int64_t* idx_out, float* score_out, void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace,
                  size_t cb_ws_bytes, void* stream) {
    using namespace aae_host;
    if (!enc || !cb) return fail(AAE_ERR_INVALID, "aae_encode_nn: null handle");
    if (col_stride < 1) return fail(AAE_ERR_INVALID, "col_stride %d < 1", col_stride);
    if (!cb_workspace || ((uintptr_t)cb_workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    if (B >= 1 && cb_ws_bytes < aae_codebook_workspace_bytes(cb, B, 1)) return fail(AAE_ERR_WORKSPACE, "codebook workspace %zu B too small", cb_ws_bytes);
    // B <= 4: the scan finishes inside its own launch; its ticket words sit at the front of the codebook workspace and
    // are prepared by the encoder's first kernel, several launches ahead on the same stream
    const aae_codebook* eff = (col_stride > 1 && cb->upright && cb->upright_stride == col_stride) ? cb->upright : cb;
    ExtraTicketPrep extra;
    const bool masked = col_stride > 1 && eff == cb;            // (no compacted copy for this stride)
    if (B >= 1 && eff->scan_ticket >= 1 && plan_scan(eff, B, 1, masked).stream) {
        extra.words = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(cb_workspace) + plan_scan(eff, B, 1, masked).ticket_off);
        extra.count = aae::kTicketSlotWords;
        extra.nonce = next_nonce();
    }
    // ... and when the whole query runs as conv1 + ONE persistent launch (detect_chain.h), the scan is that launch's last phase:
    // fp32 rows, stride 1 (the upright search on its compacted copy), answers written by the last block to arrive
    if (extra.words && B <= 4 && eff->dtype == AAE_DTYPE_F32 && (col_stride == 1 || eff != cb) && z_out && idx_out && score_out) {
        const ScanPlan sp = plan_scan(eff, B, 1);
        unsigned char* cbase = static_cast<unsigned char*>(cb_workspace);
        aae::ScanArgs& a = extra.scan;
        a.E = eff->E; a.e_bytes = (unsigned)((size_t)eff->N * eff->J * sizeof(float));
        a.q = nullptr; a.qp = nullptr; a.cs = nullptr; a.z = z_out;
        a.pval = reinterpret_cast<float*>(cbase + sp.pval_off);
        a.pidx = reinterpret_cast<int*>(cbase + sp.pidx_off);
        a.N = eff->N; a.J = eff->J; a.Jpad = sp.Jpad; a.B = B; a.Bpad = sp.Bpad; a.Bstride = sp.Bstride; a.col_stride = 1;
        a.tickets = extra.words; a.nonce = extra.nonce;
        a.idx_out = reinterpret_cast<long long*>(idx_out); a.score_out = score_out; a.idx_scale = eff != cb ? col_stride : 1;
        extra.scan_ready = true;
    }
    bool prepared = false, scan_done = false;
    Timer tm;
    if (int rc = forward_impl(enc, x, x_dtype, B, z_out, enc_workspace, enc_ws_bytes, stream, tm, extra.words ? &extra : nullptr, &prepared, &scan_done)) return rc;
    if (scan_done) return AAE_OK;
    return nn_impl(cb, z_out, B, 1, col_stride, idx_out, score_out, cb_workspace, cb_ws_bytes, stream, prepared ? extra.nonce : 0u);
}

int aae_codebook_similarity(aae_codebook* cb, const float* z, int B, float* cs_out, void* workspace, size_t ws_bytes,
                            void* stream_v) {
    using namespace aae_host;
    if (!cb || !z || !cs_out) return fail(AAE_ERR_INVALID, "aae_codebook_similarity: null argument");
    if (B < 1) return fail(AAE_ERR_INVALID, "aae_codebook_similarity: B=%d", B);
    const ScanPlan s = plan_scan(cb, B, 1);
    if (ws_bytes < s.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, s.total);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    return run_scan(cb, z, B, 1, cs_out, s, static_cast<unsigned char*>(workspace), static_cast<hipStream_t>(stream_v));
}

int aae_l2_normalize(const float* z, int B, int J, float* q_out, void* stream_v) {
    using namespace aae_host;
    if (!z || !q_out || B < 1 || J < 1) return fail(AAE_ERR_INVALID, "aae_l2_normalize: bad argument");
    aae::L2NormArgs n;
    n.z = z; n.q = q_out; n.qp = nullptr; n.B = B; n.J = J; n.Jpad = J; n.Bpad = B;
    AAE_LAUNCH((aae::l2norm_pack_kernel), dim3(ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream_v), n);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_pack_pairs(const int64_t* idx, const float* score, const int32_t* pos, int n, int stride, int64_t* packed, void* stream_v) {
    using namespace aae_host;
    if (!idx || !score || !packed) return fail(AAE_ERR_INVALID, "aae_pack_pairs: null argument");
    if (n < 0 || stride < 1) return fail(AAE_ERR_INVALID, "aae_pack_pairs: n=%d stride=%d", n, stride);
    if (n == 0) return AAE_OK;
    aae::PackPairsArgs a;
    a.idx = reinterpret_cast<const long long*>(idx); a.score = score; a.pos = pos; a.packed = reinterpret_cast<long long*>(packed);
    a.n = n; a.stride = stride;
    AAE_LAUNCH((aae::pack_pairs_kernel), dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_unpack_pairs(const int64_t* gathered, const int32_t* owner, int n, int rows_per_rank, int64_t* idx_out, float* score_out,
                     void* stream_v) {
    using namespace aae_host;
    if (!gathered || !idx_out || !score_out) return fail(AAE_ERR_INVALID, "aae_unpack_pairs: null argument");
    if (n < 0 || rows_per_rank < n) return fail(AAE_ERR_INVALID, "aae_unpack_pairs: n=%d rows_per_rank=%d", n, rows_per_rank);
    if (n == 0) return AAE_OK;
    aae::UnpackPairsArgs a;
    a.gathered = reinterpret_cast<const long long*>(gathered); a.owner = owner; a.idx = reinterpret_cast<long long*>(idx_out);
    a.score = score_out; a.n = n; a.rows_per_rank = rows_per_rank;
    AAE_LAUNCH((aae::unpack_pairs_kernel), dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_crop_resize_u8(const void* img, int H, int W, int C, const int32_t* boxes, int D, int out_h, int out_w,
                       void* out, void* stream_v) {
    using namespace aae_host;
    if (!img || !boxes || !out) return fail(AAE_ERR_INVALID, "aae_crop_resize_u8: null argument");
    if (H < 1 || W < 1 || C < 1 || D < 1 || out_h < 1 || out_w < 1)
        return fail(AAE_ERR_INVALID, "aae_crop_resize_u8: image %dx%dx%d, %d boxes, output %dx%d", H, W, C, D, out_h, out_w);
    if (D > 65535) return fail(AAE_ERR_UNSUPPORTED, "aae_crop_resize_u8: at most 65535 boxes per call");
    aae::CropResizeArgs a;
    a.img = static_cast<const unsigned char*>(img); a.boxes = boxes; a.out = static_cast<unsigned char*>(out);
    a.H = H; a.W = W; a.C = C; a.D = D; a.OH = out_h; a.OW = out_w;
    AAE_LAUNCH((aae::crop_resize_bilinear_u8_kernel), dim3(ceil_div(out_h * out_w, 256), D), dim3(256), 0,
               static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_detect_nn(aae_encoder* enc, aae_codebook* cb, const void* img, int H, int W, int C, const int32_t* boxes, int n,
                  int col_stride, void* crops, float* z_out, int64_t* idx_out, float* score_out,
                  void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace, size_t cb_ws_bytes, void* stream) {
    using namespace aae_host;
    if (!enc || !cb || !crops) return fail(AAE_ERR_INVALID, "aae_detect_nn: null argument");
    if (C != enc->desc.in_c) return fail(AAE_ERR_INVALID, "aae_detect_nn: image has %d channels, the encoder takes %d", C, enc->desc.in_c);
    if (int rc = aae_crop_resize_u8(img, H, W, C, boxes, n, enc->desc.in_h, enc->desc.in_w, crops, stream)) return rc;
    return aae_encode_nn(enc, cb, crops, AAE_DTYPE_U8, n, col_stride, z_out, idx_out, score_out, enc_workspace, enc_ws_bytes, cb_workspace,
                         cb_ws_bytes, stream);
}

}  // extern "C"
