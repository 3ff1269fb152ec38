// Layer launches and the forward pass of the encoder (/root/reference/auto_pose/ae/encoder.py:37-68: conv -> ReLU -> [BN] per layer,
// flatten, dense): one function per kernel family that fills the kernel's argument record from a Layer and the plan, then
// forward_impl, which walks the layers.  Part of aae_hip_impl.h.
#pragma once

namespace aae_host {

// ------------------------------------------------------------ layer launches
struct Timer {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
    int mark() {
        if (!on) return AAE_OK;
        hipEvent_t e;
        AAE_HIP_TRY(hipEventCreate(&e));
        ev.push_back(e);
        AAE_HIP_TRY(hipEventRecord(e, stream));
        return AAE_OK;
    }
};

static int launch_igemm(aae_encoder* enc, const Layer& L, const float* x, int M, float* out, float* partial,
                        hipStream_t stream, Timer& tm, const char* name, int tag = 0) {
    aae::ConvIgemmArgs a;
    a.x = x; a.wp = L.wp; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu; a.stagger = enc->igemm_stagger;
    const unsigned long long x_bytes = (unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * sizeof(float);
    if (x_bytes >= 0xFFFFFFF0ull)
        return fail(AAE_ERR_UNSUPPORTED, "%s: input activation of %llu bytes exceeds the 4 GiB buffer view; use a smaller batch", name, x_bytes);
    a.x_bytes = (unsigned)x_bytes;
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = ceil_div(M, 128);
    a.num_nt = L.CoutPad / 128;
    choose_splits(enc, a.num_mt * a.num_nt, a.slabs_total, &a.splits, &a.slabs_per_split);
    const int nblk = a.num_mt * a.num_nt * a.splits;
    const double flops = 2.0 * (double)M * (double)L.K() * (double)L.Cout;
#ifdef AAE_EXPERIMENTS
    const bool dma = enc->igemm_dma != 0;
#else
    const bool dma = true;                     // (the register-staged operand path: experiments build)
#endif
    const char* kname = dma ? "conv_igemm_f32_dma" : "conv_igemm_f32";
    char label[96];
    if (a.splits == 1) {
        a.out = out;
        // A buffers only (32 KB: three blocks per CU); grids too small to give every CU three blocks keep the
        // 64 KB footprint so that the blocks spread two per CU instead of 3/2/1
        const int kBregSmem = nblk >= enc->igemm_breg_min_blocks ? 2 * aae::kSlabFloatsA * 4 : aae::kConvIgemmSmem;
#ifdef AAE_EXPERIMENTS
        const bool breg = dma && enc->igemm_breg && tag >= 1 && tag <= 3;
#else
        const bool breg = tag >= 1 && tag <= 3;   // (weights through LDS for the conv layers: experiments build)
#endif
        if (breg) kname = "conv_igemm_f32_dma_breg";
        // 128 x 256 block tiles (each wave 64 x 128) where the layer is wide enough and the grid stays large
        if (breg && enc->igemm_breg_wide && (tag == 1 || tag == 2) && L.CoutPad % 256 == 0 &&
            a.num_mt * (L.CoutPad / 256) >= enc->igemm_breg_wide_min_blocks) {
            a.num_nt = L.CoutPad / 256;
            const int wide_blocks = a.num_mt * a.num_nt;
            constexpr int smem = 2 * aae::kSlabFloatsA * 4;
            if (tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1, true, 4>), dim3(wide_blocks), dim3(256), smem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2, true, 4>), dim3(wide_blocks), dim3(256), smem, stream, a);
            snprintf(label, sizeof(label), "%s:conv_igemm_f32_dma_breg_n256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
            note_kernel({label, flops});
            AAE_HIP_TRY(hipGetLastError());
            return tm.mark();
        }
        if (breg && tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
        else if (breg && tag == 2) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
        else if (breg && tag == 3) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 3, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
#ifdef AAE_EXPERIMENTS
        else if (dma && tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (dma && tag == 2) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (dma && tag == 3) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 3>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (!dma) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, false>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
#endif
        else AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        snprintf(label, sizeof(label), "%s:%s M=%d N=%d K=%lld", name, kname, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    a.out = partial;
#ifdef AAE_EXPERIMENTS
    if (!dma) AAE_LAUNCH((aae::conv_igemm_f32_kernel<true, false>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    else
#endif
    AAE_LAUNCH((aae::conv_igemm_f32_kernel<true, true>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    snprintf(label, sizeof(label), "%s:%s_splitk%d M=%d N=%d K=%lld", name, kname, a.splits, M, L.Cout, L.K());
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = L.bias; r.bn_scale = L.bn_scale; r.bn_shift = L.bn_shift; r.out = out;
    r.MN = (long long)M * L.Cout; r.Cout = L.Cout; r.splits = a.splits; r.relu = L.relu;
    r.out_planes = 0; r.out_scale = 1.f;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    snprintf(label, sizeof(label), "%s:splitk_reduce", name);
    note_kernel({label, 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// wave-split-K igemm (conv_wavek_f32.h): small M -- the per-detection batches
template <int MT, int NT, int WAVES, int DEPTH, bool SPREAD = false>
static void launch_wavek_t(const aae::ConvWaveKArgs& a, int tag, int nblk, hipStream_t stream) {
    constexpr int smem = aae::conv_wavek_smem<MT, NT, WAVES>();
    // TAG only makes the symbol unique per encoder layer (separate rows in rocprofv3 --stats)
    if (tag == 1) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 1, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 1, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else if (tag == 2) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 2, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 2, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else if (tag == 3) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 3, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 3, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 0, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 0, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    }
}

static aae::ConvWaveKArgs wavek_args(const aae_encoder* enc, const Layer& L, const WaveKPlan& w, const float* x, int M, float* out, float* partial,
                                     unsigned long long* tickets, unsigned nonce, int tag) {
    aae::ConvWaveKArgs a;
    a.x = x; a.wp = L.wp; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.out = out;
    a.partial = partial; a.partial_bytes = (unsigned)w.partial_bytes; a.tickets = tickets; a.nonce = nonce;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu; a.ablate = enc->wavek_ablate; a.pingpong = enc->wavek_pingpong; a.spread = enc->wavek_spread;
    a.timeline = (enc->wavek_timeline && tag >= 1 && tag <= 3 && w.blocks() <= 512)       // (the debug buffer holds 512 blocks per layer)
                     ? enc->wavek_timeline + (size_t)(tag - 1) * 512 * 8 : nullptr;
    a.x_bytes = (unsigned)((unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * sizeof(float));
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = w.num_mt; a.num_nt = w.num_nt; a.gsplits = w.gsplits;
    a.tail_tiles = w.tail_tiles; a.tail_gsplits = w.tail_g;
    return a;
}

static int launch_wavek(aae_encoder* enc, const Layer& L, const WaveKPlan& w, const float* x, int M, float* out, float* partial,
                        unsigned long long* tickets, unsigned nonce, hipStream_t stream, Timer& tm, const char* name, int tag) {
    const aae::ConvWaveKArgs a = wavek_args(enc, L, w, x, M, out, partial, tickets, nonce, tag);
    const int nblk = w.blocks();
    const int key = (w.MT == 1 ? 1000 : 0) + w.NT * 100 + w.waves * 10 + w.depth;
    // The product build carries the three forms the planner uses -- 4 waves, 2 slabs in flight, the spread schedules -- ; the forms
    // that measured slower (8 waves, 3 slabs in flight, the burst schedules: CHANGELOG.md rounds 2-4) exist in the experiments build.
    switch (key + ((key == 242 && !(a.spread & 1)) || (key == 1142 && !(a.spread & 2)) ? 100000 : 0)) {
        case 242: launch_wavek_t<2, 2, 4, 2, true>(a, tag, nblk, stream); break;
        case 142: launch_wavek_t<2, 1, 4, 2>(a, tag, nblk, stream); break;
        case 1142: launch_wavek_t<1, 1, 4, 2, true>(a, tag, nblk, stream); break;
#ifdef AAE_EXPERIMENTS
        case 100242: launch_wavek_t<2, 2, 4, 2>(a, tag, nblk, stream); break;
        case 101142: launch_wavek_t<1, 1, 4, 2>(a, tag, nblk, stream); break;
        case 243: launch_wavek_t<2, 2, 4, 3>(a, tag, nblk, stream); break;
        case 282: launch_wavek_t<2, 2, 8, 2>(a, tag, nblk, stream); break;
        case 143: launch_wavek_t<2, 1, 4, 3>(a, tag, nblk, stream); break;
        case 1143: launch_wavek_t<1, 1, 4, 3>(a, tag, nblk, stream); break;
        case 182: launch_wavek_t<2, 1, 8, 2>(a, tag, nblk, stream); break;
        case 1182: launch_wavek_t<1, 1, 8, 2>(a, tag, nblk, stream); break;
#endif
        default: return fail(AAE_ERR_RUNTIME, "%s: no wave-split-K instantiation for MT=%d NT=%d waves=%d depth=%d spread=%d in this build", name, w.MT, w.NT, w.waves, w.depth, a.spread);
    }
    char label[128], tail[24] = "";
    if (w.tail_tiles > 0) snprintf(tail, sizeof(tail), "t%dx%d", w.tail_tiles, w.tail_g);       // (e.g. g1t64x4: the last 64 tiles cut four ways)
    snprintf(label, sizeof(label), "%s:conv_wavek_f32_%dx%d_w%d_d%d_g%d%s M=%d N=%d K=%lld", name, 32 * w.MT, 32 * w.NT, w.waves, w.depth,
             w.gsplits, tail, M, L.Cout, L.K());
    note_kernel({label, 2.0 * (double)M * (double)L.K() * (double)L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// f32x3h variant: x and (unless out_f32) out are fp16 (hi, lo) pairs of value * 2^x3h_act_shift (x3h_pair_index layout).
static int launch_igemm_x3h(aae_encoder* enc, const Layer& L, const void* x, int M, void* out, bool out_f32, float* partial,
                            hipStream_t stream, Timer& tm, const char* name, int tag = 0) {
    aae::ConvIgemmX3hArgs a;
    a.x = static_cast<const unsigned short*>(x); a.wp = L.wp16; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu;
    const unsigned long long in_bytes = (unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * 4;
    if (in_bytes >= 0xFFFFFFF0ull)
        return fail(AAE_ERR_UNSUPPORTED, "%s: input activation of %llu bytes exceeds the 4 GiB buffer view; use a smaller batch", name, in_bytes);
    a.x_bytes = (unsigned)in_bytes;
    a.inv_scale = ldexpf(1.f, -(enc->x3h_act_shift + L.w_shift));
    a.out_scale = ldexpf(1.f, enc->x3h_act_shift);
    a.sat_flag = out_f32 ? nullptr : t_x3h_flag;
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = ceil_div(M, 128);
    a.num_nt = L.CoutPad / 128;
    choose_splits(enc, a.num_mt * a.num_nt, a.slabs_total, &a.splits, &a.slabs_per_split);
    const int nblk = a.num_mt * a.num_nt * a.splits;
    const double flops = 2.0 * (double)M * (double)L.K() * (double)L.Cout;
#ifdef AAE_EXPERIMENTS
    const bool dma = enc->x3h_dma != 0;
#else
    const bool dma = true;
#endif
    const char* kname = dma ? "conv_igemm_x3h_dma" : "conv_igemm_x3h";
    char label[96];
    // 256 x 256 tiles, 8 waves of 64 x 128 (LDS traffic per MFMA -33 %): layers with Cout % 256 == 0 whose grid still fills the chip
    if (dma && !out_f32 && enc->x3h_wide256 && L.CoutPad % 256 == 0 && tag >= 1 && tag <= 3 &&
        ceil_div(M, 256) * (L.CoutPad / 256) >= enc->x3h_wide256_min_blocks) {
        a.num_mt = ceil_div(M, 256);
        a.num_nt = L.CoutPad / 256;
        a.splits = 1;
        a.slabs_per_split = a.slabs_total;
        a.out = out;
        const int wide_blocks = a.num_mt * a.num_nt;
        constexpr int smem = aae::kX3hWideSmem;
        if (tag == 1) {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 1>), dim3(wide_blocks), dim3(512), smem, stream, a);
        } else if (tag == 2) {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 2>), dim3(wide_blocks), dim3(512), smem, stream, a);
        } else {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 3>), dim3(wide_blocks), dim3(512), smem, stream, a);
        }
        snprintf(label, sizeof(label), "%s:conv_igemm_x3h_wide256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
#ifdef AAE_EXPERIMENTS
    // 256 x 128 tiles (8 waves, one block per CU) when they still give every CU a block
    if (dma && !out_f32 && tag >= 1 && tag <= 3 && enc->x3h_wide_min_blocks > 0 &&
        ceil_div(M, 256) * a.num_nt >= enc->x3h_wide_min_blocks) {
        a.num_mt = ceil_div(M, 256);
        a.splits = 1;
        a.slabs_per_split = a.slabs_total;
        a.out = out;
        const int wide_blocks = a.num_mt * a.num_nt;
        constexpr int smem = aae::x3h_dma_smem<4>();
        if (tag == 1) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        else if (tag == 2) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        else AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        snprintf(label, sizeof(label), "%s:conv_igemm_x3h_dma256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
#endif
    if (a.splits == 1) {
        a.out = out;
        if (dma) {
            if (out_f32) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_F32>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 1) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 2) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 3) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        }
#ifdef AAE_EXPERIMENTS
        else {
            if (out_f32) AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_F32>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PLANES>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        }
#endif
        snprintf(label, sizeof(label), "%s:%s M=%d N=%d K=%lld", name, kname, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    a.out = partial;
#ifdef AAE_EXPERIMENTS
    if (!dma) AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PARTIAL>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    else
#endif
    AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PARTIAL>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    snprintf(label, sizeof(label), "%s:%s_splitk%d M=%d N=%d K=%lld", name, kname, a.splits, M, L.Cout, L.K());
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = L.bias; r.bn_scale = L.bn_scale; r.bn_shift = L.bn_shift; r.out = static_cast<float*>(out);
    r.MN = (long long)M * L.Cout; r.Cout = L.Cout; r.splits = a.splits; r.relu = L.relu;
    r.out_planes = out_f32 ? 0 : 1; r.out_scale = a.out_scale; r.sat_flag = out_f32 ? nullptr : t_x3h_flag;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    snprintf(label, sizeof(label), "%s:splitk_reduce", name);
    note_kernel({label, 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

template <int KS, int C>
static void launch_first_t(const aae::ConvFirstArgs& a, bool u8, bool planes, dim3 grid, int smem, hipStream_t stream, bool group_split = false) {
    const bool vec4 = u8 && a.vec4;
    if (group_split && !planes) {                 // per-detection batches: one block per 32-pixel group (grid.z = 4)
        const dim3 g4(grid.x, grid.y, 4);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, true, true>), g4, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, false, true>), g4, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, false, false, true>), g4, dim3(256), smem, stream, a);
        return;
    }
    if (planes) {
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, true, true>), grid, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, true>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, true>), grid, dim3(256), smem, stream, a);
    } else {
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, true>), grid, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, false>), grid, dim3(256), smem, stream, a);
    }
}

// arguments of the first-layer kernel for a batch of B crops; returns the number of tile runs (blocks along grid.x)
static int first_core_args(const aae_encoder* enc, const Layer& L, const void* x, bool u8, int B, float* out, bool planes, aae::ConvFirstCore& a) {
    a.x = x; a.lut = enc->lut; a.w = L.w_hwio; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.out = out; a.H = L.H; a.W = L.W; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout;
    a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.relu = L.relu;
    a.vec4 = (u8 && L.rowlen4 > 0 && enc->first_vec4 && (reinterpret_cast<uintptr_t>(x) & 3) == 0) ? 1 : 0;   // dword loads want a 4-byte aligned batch
    a.rowlen = a.vec4 ? L.rowlen4 : L.rowlen;
    a.lead = a.vec4 ? L.lead4 : 0;
    a.out_scale = ldexpf(1.f, enc->x3h_act_shift);
    a.sat_flag = planes ? t_x3h_flag : nullptr;
    a.tiles_per_image = ceil_div(L.Ho * L.Wo, 128);
    a.total_tiles = B * a.tiles_per_image;
    int tpb = ceil_div(a.total_tiles, enc->first_target_blocks);
    if (tpb < 1) tpb = 1;
    if (tpb > enc->first_max_tiles_per_block) tpb = enc->first_max_tiles_per_block;
    a.tiles_per_block = tpb;
    return ceil_div(a.total_tiles, tpb);
}

static int launch_first(aae_encoder* enc, const Layer& L, const void* x, bool u8, int B, float* out, bool planes,
                        hipStream_t stream, Timer& tm, const aae::TicketPrep* prep = nullptr) {
    aae::ConvFirstArgs a;
    if (prep) a.prep = *prep;
    else a.prep.n = 0;
    const int runs = first_core_args(enc, L, x, u8, B, out, planes, a);
    const dim3 grid(runs + (a.prep.n > 0 ? 1 : 0), ceil_div(L.Cout, 128));    // + the ticket-preparation block
    // per-detection batches: the four 32-pixel groups of every tile go to four blocks (10.6 -> ? us at B = 1)
    const bool group_split = !planes && a.total_tiles <= enc->first_group_split_max_tiles;
    if (L.Cin == 3) launch_first_t<5, 3>(a, u8, planes, grid, L.first_smem, stream, group_split);
    else launch_first_t<5, 1>(a, u8, planes, grid, L.first_smem, stream, group_split);
    char label[96];
    snprintf(label, sizeof(label), "conv1:conv_first_f32%s M=%d N=%d K=%lld", group_split ? "_g4" : "", B * L.Ho * L.Wo, L.Cout, L.K());
    note_kernel({label, 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

static int launch_generic(aae_encoder* enc, const Layer& L, const void* x, bool u8, long long B, float* out,
                          hipStream_t stream, Timer& tm, const char* name) {
    aae::ConvDirectArgs a;
    a.x = x; a.lut = enc->lut; a.w = L.w_hwio; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.out = out; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.relu = L.relu;
    a.total = B * L.Ho * L.Wo * L.Cout;
    long long blocks = (a.total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (u8) AAE_LAUNCH((aae::conv_direct_generic_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else AAE_LAUNCH((aae::conv_direct_generic_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    char label[96];
    snprintf(label, sizeof(label), "%s:conv_direct_generic", name);
    note_kernel({label, 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// dense layer at B <= 8: weight-streaming GEMV + the fixed-order chunk reduction
static int gemv_max_batch(const aae_encoder* enc) {
    return enc->dense_gemv_max_batch < 1 ? 1 : (enc->dense_gemv_max_batch > aae::kGemvMaxBatch ? aae::kGemvMaxBatch : enc->dense_gemv_max_batch);
}
static bool gemv_uses_ticket(const aae_encoder* enc, const Layer& D) {
    return enc->gemv_ticket && D.Cout % 4 == 0 && D.CoutPad / 128 <= kGemvTicketSlots;
}

static aae::DenseGemvArgs gemv_args(const Layer& D, const float* x, int B, float* partial) {
    aae::DenseGemvArgs a;
    a.x = x; a.wp = D.wp; a.partial = partial; a.B = B; a.K = (int)D.K(); a.Cout = D.Cout; a.CoutPad = D.CoutPad;
    a.wp_bytes = (unsigned)((unsigned long long)(D.K() / 4) * D.CoutPad * 16ull);
    a.partial_bytes = (unsigned)gemv_partial_bytes(D, B);
    a.bias = nullptr; a.bn_scale = nullptr; a.bn_shift = nullptr; a.out = nullptr; a.tickets = nullptr; a.nonce = 0; a.relu = 0;
    return a;
}

static int launch_dense_gemv(aae_encoder* enc, const Layer& D, const float* x, int B, float* out, float* partial,
                             unsigned long long* tickets, unsigned nonce, hipStream_t stream, Timer& tm) {
    aae::DenseGemvArgs a = gemv_args(D, x, B, partial);
    const int chunks = ceil_div(a.K, aae::kGemvChunk);
    const dim3 grid(chunks, D.CoutPad / 128);
    const int MQ = B <= 4 ? B : 8;
    int smem = 2 * MQ * aae::kGemvChunk * (int)sizeof(float) + 16;
    const bool ticket = tickets && gemv_uses_ticket(enc, D);
    char label[96];
    if (ticket) {
        a.bias = D.bias; a.bn_scale = D.bn_scale; a.bn_shift = D.bn_shift; a.out = out;
        a.tickets = tickets;
        a.nonce = nonce; a.relu = D.relu;
        if (smem < aae::kGemvTicketSmem) smem = aae::kGemvTicketSmem;
        if (MQ == 1) AAE_LAUNCH((aae::dense_gemv_f32_kernel<1, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 2) AAE_LAUNCH((aae::dense_gemv_f32_kernel<2, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 3) AAE_LAUNCH((aae::dense_gemv_f32_kernel<3, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 4) AAE_LAUNCH((aae::dense_gemv_f32_kernel<4, true>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::dense_gemv_f32_kernel<8, true>), grid, dim3(256), smem, stream, a);
        snprintf(label, sizeof(label), "dense:dense_gemv_f32_ticket chunks=%d M=%d N=%d K=%d", chunks, B, D.Cout, a.K);
        note_kernel({label, 2.0 * B * (double)D.K() * D.Cout});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
#ifndef AAE_EXPERIMENTS
    return fail(AAE_ERR_RUNTIME, "dense GEMV without its in-launch finish: experiments build only");      // (forward_impl never plans it here)
#else
    a.bias = nullptr; a.bn_scale = nullptr; a.bn_shift = nullptr; a.out = nullptr; a.tickets = nullptr; a.nonce = 0; a.relu = 0;
    if (MQ == 1) AAE_LAUNCH((aae::dense_gemv_f32_kernel<1>), grid, dim3(256), smem, stream, a);
    else if (MQ == 2) AAE_LAUNCH((aae::dense_gemv_f32_kernel<2>), grid, dim3(256), smem, stream, a);
    else if (MQ == 3) AAE_LAUNCH((aae::dense_gemv_f32_kernel<3>), grid, dim3(256), smem, stream, a);
    else if (MQ == 4) AAE_LAUNCH((aae::dense_gemv_f32_kernel<4>), grid, dim3(256), smem, stream, a);
    else AAE_LAUNCH((aae::dense_gemv_f32_kernel<8>), grid, dim3(256), smem, stream, a);
    snprintf(label, sizeof(label), "dense:dense_gemv_f32 chunks=%d M=%d N=%d K=%d", chunks, B, D.Cout, a.K);
    note_kernel({label, 2.0 * B * (double)D.K() * D.Cout});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = D.bias; r.bn_scale = D.bn_scale; r.bn_shift = D.bn_shift; r.out = out;
    r.MN = (long long)B * D.Cout; r.Cout = D.Cout; r.splits = chunks; r.relu = D.relu; r.out_planes = 0; r.out_scale = 1.f;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    note_kernel({"dense:splitk_reduce", 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
#endif
}

// One extra ticket range the first kernel of the forward prepares for a launch that FOLLOWS the encoder on the same
// stream (the single-launch codebook scan of aae_encode_nn).
struct ExtraTicketPrep {
    unsigned long long* words = nullptr;
    int count = 0;
    unsigned nonce = 0;
    // aae_encode_nn, B <= 4, top-1 on the fp32 stream scan: the scan itself, ready to run as the last phase of the persistent
    // per-detection launch (tickets / nonce = the words above)
    bool scan_ready = false;
    aae::ScanArgs scan;
};

#ifdef AAE_EXPERIMENTS
// The persistent per-detection launch (detect_chain.h) serves a forward when every layer behind the first runs the
// wave-split-K kernel in one of its three 4-wave / depth-2 shapes, the dense layer is the ticketed GEMV, and each layer
// output has its own buffer.
static int chain_shape_code(const WaveKPlan& w) { return w.MT == 1 ? 0 : (w.NT == 1 ? 1 : 2); }

// the instantiated (batch class, shape sequence) kernels: what plan_wavek gives the reference network at B = 1, 2, 3, 4
struct ChainVariant { int mq, s0, s1, s2; };
static const ChainVariant kChainVariants[] = {{1, 0, 0, 0}, {2, 1, 0, 0}, {4, 0, 1, 0}, {4, 2, 1, 0}};

static bool chain_eligible(const aae_encoder* enc, int B, const std::vector<WaveKPlan>& plans, bool dense_gemv_ticket) {
    const size_t nl = enc->layers.size();
    if (!enc->detect_chain || B > 4 || nl != (size_t)aae::kChainConv + 1 || enc->compact_workspace || !dense_gemv_ticket) return false;
    if (enc->wavek_ablate || (enc->wavek_timeline && !enc->chain_timeline)) return false;   // (profiling aids of the stand-alone launches)
    if (enc->wavek_spread != 3) return false;                  // (the phases are compiled with the default schedules)
    for (size_t li = 1; li < nl; ++li) {
        const WaveKPlan& w = plans[li];
        if (!w.use || w.waves != 4 || w.depth != 2 || w.tail_tiles > 0 || enc->layers[li].Cout % 4 != 0) return false;
        if (!((w.MT == 1 && w.NT == 1) || (w.MT == 2 && w.NT == 1) || (w.MT == 2 && w.NT == 2))) return false;
    }
    if (enc->dense.Cout % 4 != 0) return false;
    const int mq = B <= 2 ? B : 4;
    for (const ChainVariant& v : kChainVariants)
        if (v.mq == mq && v.s0 == chain_shape_code(plans[1]) && v.s1 == chain_shape_code(plans[2]) && v.s2 == chain_shape_code(plans[3])) return true;
    return false;
}

template <int MQ, int S0, int S1, int S2>
static void launch_chain_t(const aae::DetectChainArgs& a, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)aae::detect_chain_kernel<MQ, S0, S1, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kChainSmem);
    AAE_LAUNCH_RESIDENT((aae::detect_chain_kernel<MQ, S0, S1, S2>), dim3(grid), dim3(256), aae::kChainSmem, stream, a);
}

static int launch_detect_chain(aae_encoder* enc, int B, const std::vector<WaveKPlan>& plans, const std::vector<unsigned>& nonces, unsigned gemv_nonce,
                               unsigned barrier_nonce, const float* act0, unsigned char* base, const Workspace& ws, unsigned long long* tickets, float* z_out,
                               const ExtraTicketPrep* extra, hipStream_t stream, Timer& tm) {
    const size_t nl = enc->layers.size();
    const Layer& D = enc->dense;
    aae::DetectChainArgs a;
    memset(&a, 0, sizeof(a));
    const float* cur = act0;
    double flops = 0.0;
    for (size_t li = 1; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        const WaveKPlan& w = plans[li];
        float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
        a.conv[li - 1] = wavek_args(enc, L, w, cur, B * L.Ho * L.Wo, out, reinterpret_cast<float*>(base + ws.chain_partial_off[li]),
                                    tickets + li * kLayerTicketWords, nonces[li], 0);          // (tag 0: no per-layer stamps ...)
        // ... unless option chain_timeline = 1 + layer asks for the phase stamps of ONE conv layer, kept behind the launch's own stamps
        if (enc->chain_timeline == 1 + (int)li && enc->wavek_timeline && w.num_mt * w.num_nt * w.gsplits <= 256)
            a.conv[li - 1].timeline = enc->wavek_timeline + 256 * aae::kChainStamps;
        flops += 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout;
        cur = out;
    }
    a.dense = gemv_args(D, cur, B, reinterpret_cast<float*>(base + ws.chain_partial_off[nl]));
    a.dense.bias = D.bias; a.dense.bn_scale = D.bn_scale; a.dense.bn_shift = D.bn_shift; a.dense.out = z_out; a.dense.relu = D.relu;
    a.dense.tickets = tickets + kConvTicketBytes / 8; a.dense.nonce = gemv_nonce;
    a.dense_tiles = D.CoutPad / 128;
    a.dense_chunks = ceil_div((int)D.K(), aae::kGemvChunk);
    flops += 2.0 * B * (double)D.K() * D.Cout;
    a.has_scan = (extra && extra->scan_ready) ? 1 : 0;
    if (a.has_scan) {
        a.scan = extra->scan;
        flops += 2.0 * B * (double)a.scan.N * a.scan.J;
    }
    a.barrier.words = tickets + (kConvTicketBytes + kGemvTicketBytes) / 8;
    a.barrier.nonce = barrier_nonce;
    a.timeline = (enc->chain_timeline && enc->wavek_timeline) ? enc->wavek_timeline : nullptr;      // (3 * 512 * 8 stamps: up to 307 blocks x 40)
    if (a.timeline && (size_t)std::min(enc->detect_chain_blocks, enc->cu_count > 0 ? enc->cu_count : enc->detect_chain_blocks) * aae::kChainStamps > 3u * 512u * 8u) a.timeline = nullptr;
    int grid = enc->detect_chain_blocks;
    if (enc->cu_count > 0 && grid > enc->cu_count) grid = enc->cu_count;
    if (grid < 1) grid = 1;
    const int key = (B <= 2 ? B : 4) * 1000 + chain_shape_code(plans[1]) * 100 + chain_shape_code(plans[2]) * 10 + chain_shape_code(plans[3]);
    switch (key) {                                              // (kChainVariants)
        case 1000: launch_chain_t<1, 0, 0, 0>(a, grid, stream); break;
        case 2100: launch_chain_t<2, 1, 0, 0>(a, grid, stream); break;
        case 4010: launch_chain_t<4, 0, 1, 0>(a, grid, stream); break;
        case 4210: launch_chain_t<4, 2, 1, 0>(a, grid, stream); break;
        default: return fail(AAE_ERR_RUNTIME, "no persistent per-detection kernel for batch %d / wave-tile shapes %d", B, key % 1000);
    }
    char label[128];
    snprintf(label, sizeof(label), "chain:detect_chain_f32 B=%d blocks=%d shapes=%d%d%d phases=conv2..conv%zu+dense%s", B, grid, chain_shape_code(plans[1]),
             chain_shape_code(plans[2]), chain_shape_code(plans[3]), nl, a.has_scan ? "+scan" : "");
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

#endif  // AAE_EXPERIMENTS

#ifdef AAE_EXPERIMENTS
// A conv layer as four polyphase Winograd launches (conv_winograd_f32.h): the phases add up in the output buffer, the last one applies
// bias / ReLU / BN.  The 3 x 3-tap phase goes first (it stores), the 2 x 2-tap one last.
template <int GEOM>
static void launch_wino_phase(const aae::ConvWinoArgs& a, int eh, int ew, unsigned grid, hipStream_t stream) {
    constexpr int smem = aae::wino_smem_bytes<GEOM>();
    if (eh && ew) AAE_LAUNCH((aae::conv_wino_phase_kernel<3, 3, false, GEOM>), dim3(grid), dim3(512), smem, stream, a);
    else if (eh) AAE_LAUNCH((aae::conv_wino_phase_kernel<3, 2, false, GEOM>), dim3(grid), dim3(512), smem, stream, a);
    else if (ew) AAE_LAUNCH((aae::conv_wino_phase_kernel<3, 2, true, GEOM>), dim3(grid), dim3(512), smem, stream, a);
    else AAE_LAUNCH((aae::conv_wino_phase_kernel<2, 2, false, GEOM>), dim3(grid), dim3(512), smem, stream, a);
}
#endif

// 64-column blocks of a region per XCD (conv_winograd_f32.h: wino_block).  Option winograd_xcd_cols: -1 = the measured default per layer,
// 0 = plain block order, S > 0 = S where the layer's column-block count allows it.
static int wino_xcd_cols(const aae_encoder* enc, const Layer& L) {
    const int nbn = L.Cout / 64;
    int s = enc->winograd_xcd_cols;
    if (s < 0) s = nbn >= 8 ? 4 : nbn;
    while (s > 0 && !aae::wino_xcd_cols_valid(nbn, s)) s >>= 1;
    return s;
}

static int launch_winograd(aae_encoder* enc, const Layer& L, const float* x, int B, float* out, hipStream_t stream, Timer& tm, const char* name) {
    aae::ConvWinoArgs a;
    a.x = x; a.U = nullptr; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.out = out;
    a.B = B; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout; a.Ho = L.Ho; a.Wo = L.Wo; a.relu = L.relu;
    a.eh = a.ew = 0; a.mode = 0;
    a.blocks_x = L.wino_geom == 0 ? L.Wo / 16 : 1;
    a.blocks_y = L.wino_geom == 0 ? L.Ho / 16 : 1;
    a.regions = L.wino_geom == 0 ? a.blocks_x * a.blocks_y * B : ceil_div(B, 4);
    a.xcd_cols = wino_xcd_cols(enc, L);
    const unsigned grid = aae::wino_grid_blocks(a.regions, L.Cout / 64, a.xcd_cols);
    const double tiles = (double)B * (L.Ho / 2) * (L.Wo / 2);
    char label[128];
#ifdef AAE_EXPERIMENTS
    if (enc->winograd == 2) {
        // one launch per phase, the phases add up in the output buffer (A/B of the one-launch form: 2 % slower)
        static const int order[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
        for (int i = 0; i < 4; ++i) {
            const int eh = order[i][0], ew = order[i][1];
            a.eh = eh; a.ew = ew; a.U = L.wino[2 * eh + ew];
            a.mode = i == 0 ? 0 : (i == 3 ? 2 : 1);
            if (L.wino_geom == 0) launch_wino_phase<0>(a, eh, ew, grid, stream);
            else launch_wino_phase<1>(a, eh, ew, grid, stream);
            const int taps = (eh ? 3 : 2) * (ew ? 3 : 2), points = ((eh ? 3 : 2) + 1) * ((ew ? 3 : 2) + 1);
            snprintf(label, sizeof(label), "%s:conv_wino_f32 phase %d%d (%d taps as %d products per 2x2 outputs) M=%d N=%d C=%d", name, eh, ew, taps, points,
                     B * L.Ho * L.Wo, L.Cout, L.Cin);
            note_kernel({label, 2.0 * tiles * points * (double)L.Cin * (double)L.Cout});
            AAE_HIP_TRY(hipGetLastError());
            if (int rc = tm.mark()) return rc;
        }
        return AAE_OK;
    }
#endif
    // one launch for the layer: the four phases inside the block, output written once
    aae::ConvWinoLayerArgs p;
    p.c = a;
    for (int i = 0; i < 4; ++i) p.U4[i] = L.wino[i];
    wino_layer_launch(L.wino_geom, enc->winograd_wide, grid, stream, p);
    // (the record carries the flops the kernel EXECUTES -- 49 products per 2 x 2 outputs and channel pair where the direct form
    //  multiplies 100 -- so that its TFLOP/s figure is a statement about the kernel)
    snprintf(label, sizeof(label), "%s:conv_wino_f32 layer (25 taps as 49 products per 2x2 outputs) M=%d N=%d C=%d", name, B * L.Ho * L.Wo, L.Cout, L.Cin);
    note_kernel({label, 2.0 * tiles * 49.0 * (double)L.Cin * (double)L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

static bool winograd_rule(const aae_encoder* enc, const Layer& L, int B);
static bool runs_winograd(const aae_encoder* enc, const Layer& L, int B) { return L.wino[0] && winograd_rule(enc, L, B); }
// would any conv layer of a batch of B take the Winograd form, were its weights there?
static bool wants_winograd_weights(const aae_encoder* enc, int B) {
    for (const Layer& L : enc->layers)
        if (L.kind == KIND_IGEMM && !L.wino[0] && winograd_rule(enc, L, B)) return true;
    return false;
}
// the Winograd-domain weights of every eligible layer: packed from the device copy of the HWIO kernel and uploaded, once (NOT from a hot call:
// it allocates and copies synchronously -- aae_encoder_workspace_bytes / aae_multi_workspace_bytes call it for batches that will use them)
static int ensure_winograd_weights(aae_encoder* enc) {
    std::lock_guard<std::mutex> lk(enc->wino_mu);
    for (Layer& L : enc->layers) {
        if (L.kind != KIND_IGEMM || L.wino_geom < 0 || L.wino[0] || !L.w_hwio) continue;
        std::vector<float> k((size_t)L.K() * L.Cout);
        AAE_HIP_TRY(hipMemcpy(k.data(), L.w_hwio, k.size() * sizeof(float), hipMemcpyDeviceToHost));
        float* dev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int eh = 0; eh < 2; ++eh)
            for (int ew = 0; ew < 2; ++ew) {
                const std::vector<float> U = pack_weights_winograd(k.data(), L.KS, L.Cin, L.Cout, eh, ew, !eh && ew);
                if (int rc = upload(enc, U.data(), U.size(), &dev[2 * eh + ew])) return rc;
            }
        for (int q = 3; q >= 0; --q) L.wino[q] = dev[q];          // (component 0 last: runs_winograd asks for it)
    }
    return AAE_OK;
}
static bool winograd_rule(const aae_encoder* enc, const Layer& L, int B) {
    if (!enc->winograd || L.wino_geom < 0 || B < enc->winograd_min_batch) return false;
    // (the kernel reads the input through a 32-bit buffer view whose upper half marks "outside the image": activations below 2 GiB)
    const unsigned long long x_bytes = (unsigned long long)B * L.H * L.W * L.Cin * sizeof(float);
    if (x_bytes >= 0x7FFFFF00ull) return false;
    // A launch costs WHOLE rounds of blocks (one block per compute unit at a time: 121-131 KB of LDS), and a block-slot of the Winograd form
    // costs 0.55-0.6 of what the direct kernels need for the same outputs (tools/wino_ab.py, profiles/r15/winograd_vs_direct_every_layer_forced.jsonl:
    // direct / Winograd = 1.65-1.85 at full rounds, 1.0 where the blocks fill half of the rounds they occupy): a layer gains when its blocks
    // fill at least winograd_min_fill_pct (56) per cent of their rounds -- default net: conv2 from B = 9, conv3 from 18, conv4 (four images
    // per block) from 69, and e.g. not conv3 at B = 33 ... 35 or conv4 at B = 129 ... 140 (a new round for a few blocks)
    const long long blocks = (long long)(L.Cout / 64) * (L.wino_geom == 0 ? (long long)(L.Ho / 16) * (L.Wo / 16) * B : (long long)ceil_div(B, 4));
    if (enc->winograd_min_blocks > 0) return blocks >= enc->winograd_min_blocks;
    const long long cus = wavek_round_blocks(enc), rounds = (blocks + cus - 1) / cus;
    return 100 * blocks >= (long long)enc->winograd_min_fill_pct * rounds * cus;
}

// layer_begin / layer_end: run only the layers [layer_begin, layer_end) of the chain -- conv layers 0 ... nl - 1, the dense layer = nl; x is
// then the INPUT of layer layer_begin (fp32 activation when layer_begin > 0) and the workspace the one of a whole forward (the grouped
// mid-batch query runs conv1 and the dense layer per object around its one-launch-per-layer Winograd convolutions: aae_multi_impl.h).
static int forward_impl(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                        size_t ws_bytes, void* stream_v, Timer& tm, const ExtraTicketPrep* extra = nullptr, bool* extra_prepared = nullptr,
                        bool* scan_done = nullptr, int layer_begin = 0, int layer_end = -1) {
    if (extra_prepared) *extra_prepared = false;
    if (scan_done) *scan_done = false;
    if (!enc || !x || !z_out) return fail(AAE_ERR_INVALID, "aae_encoder_forward: null argument");
    if (B < 1) return fail(AAE_ERR_INVALID, "aae_encoder_forward: batch %d < 1", B);
    if (x_dtype != AAE_DTYPE_U8 && x_dtype != AAE_DTYPE_F32)
        return fail(AAE_ERR_INVALID, "aae_encoder_forward: x_dtype %d (want AAE_DTYPE_U8 or AAE_DTYPE_F32)", x_dtype);
    const Workspace ws = plan_workspace(enc, B);
    if (ws_bytes < ws.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B for batch %d", ws_bytes, ws.total, B);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    float* partial = reinterpret_cast<float*>(base + ws.partial_off);
    unsigned long long* tickets = reinterpret_cast<unsigned long long*>(base + ws.ticket_off);
    unsigned long long* gemv_tickets = tickets + kConvTicketBytes / 8;
    auto layer_tickets = [&](size_t li) { return tickets + li * kLayerTicketWords; };     // li == layers.size(): the dense layer
    RecordScope rec(enc);
    tm.stream = stream;
    if (int rc = tm.mark()) return rc;

    const void* cur = x;
    bool cur_u8 = (x_dtype == AAE_DTYPE_U8);
    t_x3h_last_slot = -1;
    const bool ranged = layer_begin != 0 || layer_end >= 0;
    if (layer_end < 0) layer_end = (int)enc->layers.size() + 1;
    if (ranged && (layer_begin < 0 || layer_begin >= layer_end || layer_end > (int)enc->layers.size() + 1 || runs_split(enc, B) || (layer_begin > 0 && cur_u8)))
        return fail(AAE_ERR_INVALID, "aae_encoder_forward: layer range [%d, %d) of a %zu-layer encoder", layer_begin, layer_end, enc->layers.size() + 1);
    if (runs_split(enc, B)) {
        // this forward's range flag: a ring slot, or -- while the stream is being captured into a graph -- a slot of its own
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &capture);
        int slot;
        if (capture != hipStreamCaptureStatusNone) {
            std::lock_guard<std::mutex> lk(enc->x3h_mu);
            slot = -1;
            for (size_t f = 0; f < enc->x3h_free.size(); ++f) {          // a released slot whose flag clear has executed (aae_encoder_x3h_release_slot)
                const size_t k = (size_t)(enc->x3h_free[f] - kX3hRing);
                if (k < enc->x3h_release_ev.size() && enc->x3h_release_ev[k] && hipEventQuery(enc->x3h_release_ev[k]) != hipSuccess) continue;
                slot = enc->x3h_free[f];
                enc->x3h_free.erase(enc->x3h_free.begin() + (long)f);
                break;
            }
            if (slot < 0) {
                if (enc->x3h_captured >= kX3hCaptured)         // (nothing is consumed by the failing call)
                    return fail(AAE_ERR_UNSUPPORTED, "more than %d f32x3h forwards live in HIP graphs on one encoder handle (aae_encoder_x3h_release_slot returns a destroyed graph's slot)", kX3hCaptured);
                slot = kX3hRing + enc->x3h_captured++;
            }
        } else {
            slot = (int)(enc->x3h_seq.fetch_add(1, std::memory_order_relaxed) % kX3hRing);
        }
        t_x3h_flag = enc->x3h_sat + slot;
        t_x3h_last_slot = slot;
        // f32x3h: conv1 (fp32 MFMA, K = 75) emits fp16 hi/lo planes, every later layer runs the
        // split-precision igemm on planes; only the latent z comes back as fp32.
        for (size_t li = 0; li < enc->layers.size(); ++li) {
            const Layer& L = enc->layers[li];
            float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
            char name[16];
            snprintf(name, sizeof(name), "conv%zu", li + 1);
            int rc;
            if (li == 0) rc = launch_first(enc, L, cur, cur_u8, B, out, true, stream, tm);
            else rc = launch_igemm_x3h(enc, L, cur, B * L.Ho * L.Wo, out, false, partial, stream, tm, name, (int)li);
            if (rc) return rc;
            cur = out;
        }
        return launch_igemm_x3h(enc, enc->dense, cur, B, z_out, true, partial, stream, tm, "dense");
    }

    // ---- plan every layer first: the ticketed launches get their nonces now, so that the first kernel can install them
    const size_t nl = enc->layers.size();
    const Layer& D = enc->dense;
    std::vector<WaveKPlan> plans(nl + 1);
    std::vector<unsigned> nonces(nl + 1, 0u);
    for (size_t li = 0; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        const bool first_mfma = li == 0 && L.kind == KIND_FIRST_MFMA;
        if (!first_mfma && L.kind == KIND_IGEMM && !(li == 0 && cur_u8) && !runs_winograd(enc, L, B)) plans[li] = plan_wavek(enc, L, (long long)B * L.Ho * L.Wo, false);
    }
#ifdef AAE_EXPERIMENTS
    const bool dense_gemv = D.kind == KIND_IGEMM && B <= gemv_max_batch(enc) && enc->dense_gemv && D.K() % aae::kGemvChunk == 0;
#else
    // (the GEMV whose chunk rows a second launch adds up -- latent sizes that are no multiple of 4 or beyond 1024, option
    //  gemv_ticket = 0 -- lives in the experiments build; here such a dense layer takes the wave-split-K / split-K matrix path)
    const bool dense_gemv = D.kind == KIND_IGEMM && B <= gemv_max_batch(enc) && enc->dense_gemv && D.K() % aae::kGemvChunk == 0 && gemv_uses_ticket(enc, D);
#endif
    const bool gemv_ticket = dense_gemv && gemv_uses_ticket(enc, D);
    if (!dense_gemv && D.kind == KIND_IGEMM && enc->wavek_dense) plans[nl] = plan_wavek(enc, D, B, false);
    aae::TicketPrep prep;
    prep.n = 0;
    auto add_prep = [&](unsigned long long* words, int count, unsigned nonce) {
        if (prep.n >= aae::kMaxTicketPrep) return false;
        prep.words[prep.n] = words; prep.count[prep.n] = count; prep.nonce[prep.n] = nonce; ++prep.n;
        return true;
    };
    for (size_t li = 0; li <= nl; ++li)
        if (plans[li].use && (plans[li].gsplits > 1 || plans[li].tail_tiles > 0)) {
            nonces[li] = next_nonce();
            add_prep(layer_tickets(li), plans[li].tail_tiles > 0 ? plans[li].tail_tiles : plans[li].num_mt * plans[li].num_nt, nonces[li]);
        }
    unsigned gemv_nonce = 0;
    if (gemv_ticket) {
        gemv_nonce = next_nonce();
        add_prep(gemv_tickets, (D.CoutPad / 128) * aae::kTicketSlotWords, gemv_nonce);
    }
    const bool extra_listed = extra && extra->words && add_prep(extra->words, extra->count, extra->nonce);
    // the persistent per-detection launch: its grid-barrier words count monotonically inside a launch and must start from
    // (nonce, 0) -- also when a captured graph replays the launch with the SAME nonce.  The first kernel resets them with the
    // other ticket words; where it cannot, a memset in front of the launch does.
#ifdef AAE_EXPERIMENTS
    const bool chain = !ranged && chain_eligible(enc, B, plans, gemv_ticket);
    unsigned long long* barrier_words = tickets + (kConvTicketBytes + kGemvTicketBytes) / 8;
    const unsigned barrier_nonce = chain ? next_nonce() : 0u;
    const bool barrier_listed = chain && add_prep(barrier_words, aae::kGridBarrierWords, barrier_nonce);
#endif
    const bool can_prepare = enc->ticket_prep && enc->layers[0].kind == KIND_FIRST_MFMA && prep.n > 0;
    if (extra_prepared) *extra_prepared = can_prepare && extra_listed;

#ifdef AAE_EXPERIMENTS
    // ---- per-detection batches: the first layer as its own launch, everything behind it in ONE persistent launch
    if (chain) {
        const Layer& L0 = enc->layers[0];
        float* out0 = reinterpret_cast<float*>(base + ws.act_off[0]);
        int rc;
        if (L0.kind == KIND_FIRST_MFMA) rc = launch_first(enc, L0, cur, cur_u8, B, out0, false, stream, tm, can_prepare ? &prep : nullptr);
        else if (L0.kind == KIND_IGEMM && !cur_u8) {
            if (plans[0].use)
                rc = launch_wavek(enc, L0, plans[0], static_cast<const float*>(cur), B * L0.Ho * L0.Wo, out0, partial, layer_tickets(0), nonces[0],
                                  stream, tm, "conv1", 0);
            else rc = launch_igemm(enc, L0, static_cast<const float*>(cur), B * L0.Ho * L0.Wo, out0, partial, stream, tm, "conv1", 0);
        } else rc = launch_generic(enc, L0, cur, cur_u8, B, out0, stream, tm, "conv1");
        if (rc) return rc;
        if (!(can_prepare && barrier_listed)) AAE_HIP_TRY(hipMemsetAsync(barrier_words, 0, (size_t)aae::kGridBarrierWords * 8, stream));
        rc = launch_detect_chain(enc, B, plans, nonces, gemv_nonce, barrier_nonce, out0, base, ws, tickets, z_out, extra, stream, tm);
        if (rc == AAE_OK && scan_done) *scan_done = extra && extra->scan_ready;
        return rc;
    }
#endif

    for (size_t li = (size_t)layer_begin; li < nl && (int)li < layer_end; ++li) {
        const Layer& L = enc->layers[li];
        float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
        char name[16];
        snprintf(name, sizeof(name), "conv%zu", li + 1);
        int rc;
        if (li == 0 && L.kind == KIND_FIRST_MFMA) rc = launch_first(enc, L, cur, cur_u8, B, out, false, stream, tm, can_prepare ? &prep : nullptr);
        else if (L.kind == KIND_IGEMM && !cur_u8) {
            if (runs_winograd(enc, L, B)) rc = launch_winograd(enc, L, static_cast<const float*>(cur), B, out, stream, tm, name);
            else if (plans[li].use)
                rc = launch_wavek(enc, L, plans[li], static_cast<const float*>(cur), B * L.Ho * L.Wo, out, partial, layer_tickets(li), nonces[li],
                                  stream, tm, name, (int)li);
            else rc = launch_igemm(enc, L, static_cast<const float*>(cur), B * L.Ho * L.Wo, out, partial, stream, tm, name, (int)li);
        } else rc = launch_generic(enc, L, cur, cur_u8, B, out, stream, tm, name);
        if (rc) return rc;
        cur = out;
        cur_u8 = false;
    }
    if (layer_end <= (int)nl) return AAE_OK;
    if (dense_gemv)
        return launch_dense_gemv(enc, D, static_cast<const float*>(cur), B, z_out, partial, gemv_ticket ? gemv_tickets : nullptr, gemv_nonce, stream, tm);
    if (plans[nl].use)
        return launch_wavek(enc, D, plans[nl], static_cast<const float*>(cur), B, z_out, partial, layer_tickets(nl), nonces[nl], stream, tm, "dense", 0);
    if (D.kind == KIND_IGEMM) return launch_igemm(enc, D, static_cast<const float*>(cur), B, z_out, partial, stream, tm, "dense");
    return launch_generic(enc, D, cur, false, B, z_out, stream, tm, "dense");
}


}  // namespace aae_host
