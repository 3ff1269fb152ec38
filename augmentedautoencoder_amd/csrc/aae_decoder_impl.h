// Host side of the decoder entry points (include/aae_hip.h, "next" row N4): weight folding for
// the phase-decomposed upsample+conv stages, launch planning, C ABI.  Included at the end of
// aae_hip_impl.h; uses its Layer / Timer / launch_igemm machinery with an embedded aae_encoder
// as the context that owns device allocations, launch options and kernel records.
#pragma once

#include "kernels/decoder_f32.h"

namespace aae_host {

enum StageMode { STAGE_IGEMM_PLAIN = 0, STAGE_IGEMM_PHASED = 1, STAGE_NARROW_PHASED = 2, STAGE_DIRECT = 3 };

struct DecStage {
    Layer L;                 // H, W, Cin: source resolution; Cout; KS: the reference's kernel size
    int UH = 0, UW = 0;      // resize target = output resolution of the stage
    int pad = 0;             // (KS-1)/2 of the 'same' stride-1 convolution
    int act = aae::ACT_RELU;
    StageMode mode = STAGE_DIRECT;
    // phased forms (exact 2x upsampling)
    int U = 0;               // taps per axis of a phase problem
    int ph_pt[2] = {0, 0};   // rows/cols of padding in front, per output parity
    float* wp_phases = nullptr;      // [4][U*U*Cin/4][CoutPad][4]
    long long wp_phase_floats = 0;
    float* wq = nullptr;             // narrow: [Un*Un][Cin][4][4]
    float* bias4 = nullptr;
    int Un = 0, dmin = 0;            // narrow: union of the phase tap ranges
    double nominal_flops_per_image = 0.0;
};

}  // namespace aae_host

struct aae_decoder {
    aae_decoder_desc desc;
    aae_encoder ctx;                       // allocations, options, records
    aae_host::Layer dense;
    std::vector<aae_host::DecStage> stages;
    int h0 = 0, w0 = 0;
};

namespace aae_host {

// d(k) = floor((p + k - pad) / 2): source offset that tap k of output parity p reads
static inline int phase_src_offset(int p, int k, int pad) {
    const int v = p + k - pad;
    return v >= 0 ? v / 2 : -((-v + 1) / 2);
}

// W [KS][KS][Cin][Cout] -> per phase (py,px) the folded [U][U][Cin][Cout] kernel (float64 sums).
static std::vector<float> fold_phase(const float* w, int KS, int pad, int Cin, int Cout, int py, int px, int U) {
    std::vector<double> acc((size_t)U * U * Cin * Cout, 0.0);
    const int dy0 = phase_src_offset(py, 0, pad), dx0 = phase_src_offset(px, 0, pad);
    for (int kh = 0; kh < KS; ++kh)
        for (int kw = 0; kw < KS; ++kw) {
            const int ty = phase_src_offset(py, kh, pad) - dy0, tx = phase_src_offset(px, kw, pad) - dx0;
            const float* src = w + ((size_t)(kh * KS + kw) * Cin) * Cout;
            double* dst = acc.data() + ((size_t)(ty * U + tx) * Cin) * Cout;
            for (size_t i = 0; i < (size_t)Cin * Cout; ++i) dst[i] += (double)src[i];
        }
    std::vector<float> out(acc.size());
    for (size_t i = 0; i < acc.size(); ++i) out[i] = (float)acc[i];
    return out;
}

static int upload_bn(aae_encoder* ctx, const void* const* hw, int& wi, int C, float eps, Layer& L) {
    const float* g = static_cast<const float*>(hw[wi++]);
    const float* be = static_cast<const float*>(hw[wi++]);
    const float* mu = static_cast<const float*>(hw[wi++]);
    const float* var = static_cast<const float*>(hw[wi++]);
    std::vector<float> sc(C), sh(C);
    for (int c = 0; c < C; ++c) {
        const float inv = (1.0f / sqrtf(var[c] + eps)) * g[c];
        sc[c] = inv;
        sh[c] = be[c] - mu[c] * inv;
    }
    if (int rc = upload(ctx, sc.data(), C, &L.bn_scale)) return rc;
    return upload(ctx, sh.data(), C, &L.bn_shift);
}

struct DecWorkspace {
    std::vector<size_t> act_off;       // [0] dense output, [i] output of hidden stage i-1
    std::vector<size_t> act_count;
    size_t partial_off = 0, total = 0;
};

static DecWorkspace plan_decoder_workspace(const aae_decoder* dec, int B) {
    DecWorkspace ws;
    size_t off = 0;
    auto add = [&](size_t count) {
        ws.act_off.push_back(off);
        ws.act_count.push_back(count);
        off += align_up(count * sizeof(float), 256);
    };
    add((size_t)B * dec->dense.Cout);
    for (size_t i = 0; i + 1 < dec->stages.size(); ++i) {
        const DecStage& s = dec->stages[i];
        add((size_t)B * s.UH * s.UW * s.L.Cout);
    }
    ws.partial_off = off;
    size_t partial = 0;
    if (dec->dense.kind == KIND_IGEMM) {
        int splits, per;
        choose_splits(&dec->ctx, ceil_div(B, 128) * (dec->dense.CoutPad / 128), (int)(dec->dense.K() / 32), &splits, &per);
        if (splits > 1) partial = (size_t)splits * B * dec->dense.Cout * sizeof(float);
    }
    for (const DecStage& s : dec->stages)
        if (s.mode == STAGE_IGEMM_PLAIN) {
            int splits, per;
            const int M = B * s.UH * s.UW;
            choose_splits(&dec->ctx, ceil_div(M, 128) * (s.L.CoutPad / 128), (int)(s.L.K() / 32), &splits, &per);
            if (splits > 1) partial = std::max(partial, (size_t)splits * M * s.L.Cout * sizeof(float));
        }
    off += align_up(partial, 256);
    ws.total = off;
    return ws;
}

static int launch_stage(aae_decoder* dec, const DecStage& s, const float* x, int B, float* out, float* partial,
                        hipStream_t stream, Timer& tm, const char* name) {
    aae_encoder* ctx = &dec->ctx;
    const Layer& L = s.L;
    const double flops = s.nominal_flops_per_image * B;
    char label[128];
    if (s.mode == STAGE_IGEMM_PLAIN) return launch_igemm(ctx, L, x, B * s.UH * s.UW, out, partial, stream, tm, name);
    if (s.mode == STAGE_IGEMM_PHASED) {
        aae::ConvIgemmArgs a;
        a.x = x; a.wp = s.wp_phases; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.out = out;
        a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.H; a.Wo = L.W; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
        a.KS = s.U; a.S = 1; a.pt = 0; a.pl = 0; a.M = B * L.H * L.W; a.relu = (s.act == aae::ACT_RELU); a.stagger = 0;
        const unsigned long long x_bytes = (unsigned long long)B * L.H * L.W * L.Cin * sizeof(float);
        if (x_bytes >= 0xFFFFFFF0ull) return fail(AAE_ERR_UNSUPPORTED, "%s: activation of %llu bytes exceeds the 4 GiB buffer view", name, x_bytes);
        a.x_bytes = (unsigned)x_bytes;
        a.slabs_total = s.U * s.U * L.Cin / 32;
        a.slabs_per_split = a.slabs_total;
        a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
        a.wp_phase_floats = s.wp_phase_floats;
        a.ph_pt[0] = s.ph_pt[0]; a.ph_pt[1] = s.ph_pt[1]; a.ph_pl[0] = s.ph_pt[0]; a.ph_pl[1] = s.ph_pt[1];
        a.num_mt = ceil_div(a.M, 128);
        a.num_nt = L.CoutPad / 128;
        a.splits = 4;                              // the four output phases
        const int nblk = a.num_mt * a.num_nt * 4;
        AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, true>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        snprintf(label, sizeof(label), "%s:upconv2x_igemm_f32_dma 4 phases x (M=%d N=%d K=%d)", name, a.M, L.Cout, s.U * s.U * L.Cin);
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    if (s.mode == STAGE_NARROW_PHASED) {
        aae::UpconvNarrowArgs a;
        a.x = x; a.wq = s.wq; a.bias = s.bias4; a.out = out;
        a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout; a.U = s.Un; a.dmin = s.dmin; a.act = s.act;
        a.tiles_y = ceil_div(L.H, 16); a.tiles_x = ceil_div(L.W, 16);
        a.row_stride = aae::narrow_row_stride(s.Un);
        const int smem = aae::narrow_smem_bytes(s.Un);
        (void)hipFuncSetAttribute((const void*)aae::upconv2x_narrow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::upconv2x_narrow_kernel), dim3((unsigned)(B * a.tiles_y * a.tiles_x)), dim3(256), smem, stream, a);
        snprintf(label, sizeof(label), "%s:upconv2x_narrow px=%lld Cin=%d Cout=%d taps=%dx%d", name, (long long)B * L.H * L.W, L.Cin, L.Cout, s.Un, s.Un);
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    aae::UpconvDirectArgs a;
    a.x = x; a.w = L.w_hwio; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.out = out;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.UH = s.UH; a.UW = s.UW; a.Cout = L.Cout; a.KS = L.KS; a.pad = s.pad; a.act = s.act;
    a.total = (long long)B * s.UH * s.UW * L.Cout;
    long long blocks = (a.total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    AAE_LAUNCH((aae::upconv_direct_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    snprintf(label, sizeof(label), "%s:upconv_direct", name);
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

static int decoder_forward_impl(aae_decoder* dec, const float* z, int B, float* x_out, void* workspace, size_t ws_bytes,
                                void* stream_v, Timer& tm) {
    if (!dec || !z || !x_out) return fail(AAE_ERR_INVALID, "aae_decoder_forward: null argument");
    if (B < 1) return fail(AAE_ERR_INVALID, "aae_decoder_forward: batch %d < 1", B);
    const DecWorkspace ws = plan_decoder_workspace(dec, B);
    if (ws_bytes < ws.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B for batch %d", ws_bytes, ws.total, B);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    float* partial = reinterpret_cast<float*>(base + ws.partial_off);
    aae_encoder* ctx = &dec->ctx;
    RecordScope rec(ctx);
    tm.stream = stream;
    if (int rc = tm.mark()) return rc;

    float* cur = reinterpret_cast<float*>(base + ws.act_off[0]);
    int rc;
    if (dec->dense.kind == KIND_IGEMM) rc = launch_igemm(ctx, dec->dense, z, B, cur, partial, stream, tm, "dense");
    else rc = launch_generic(ctx, dec->dense, z, false, B, cur, stream, tm, "dense");
    if (rc) return rc;
    for (size_t i = 0; i < dec->stages.size(); ++i) {
        const bool last = i + 1 == dec->stages.size();
        float* out = last ? x_out : reinterpret_cast<float*>(base + ws.act_off[i + 1]);
        char name[16];
        snprintf(name, sizeof(name), "up%zu", i + 1);
        if ((rc = launch_stage(dec, dec->stages[i], cur, B, out, partial, stream, tm, name))) return rc;
        cur = out;
    }
    return AAE_OK;
}

}  // namespace aae_host

extern "C" {

int aae_decoder_create(const aae_decoder_desc* d, const void* const* hw, int n_weights, aae_decoder** out) {
    using namespace aae_host;
    if (!d || !hw || !out) return fail(AAE_ERR_INVALID, "aae_decoder_create: null argument");
    if (d->num_layers < 1 || d->num_layers > AAE_MAX_LAYERS) return fail(AAE_ERR_INVALID, "num_layers %d outside [1,%d]", d->num_layers, AAE_MAX_LAYERS);
    if (d->out_h < 1 || d->out_w < 1 || d->out_c < 1 || d->kernel_size < 1 || d->latent_size < 1 || (d->kernel_size & 1) == 0)
        return fail(AAE_ERR_INVALID, "decoder desc: non-positive shape or even kernel size %d", d->kernel_size);
    const int L = d->num_layers;
    const int bn = d->batch_norm ? 4 : 0;
    const int expect = (2 + bn) + (L - 1) * (2 + bn) + 2;
    if (n_weights != expect) return fail(AAE_ERR_INVALID, "expected %d weight arrays, got %d", expect, n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!hw[i]) return fail(AAE_ERR_INVALID, "weight array %d is null", i);
    // layer_dimensions[i] = int(h / prod(strides[i:]))   (decoder.py:41)
    std::vector<int> dh(L), dw(L);
    for (int i = 0; i < L; ++i) {
        long long prod = 1;
        for (int j = i; j < L; ++j) {
            if (d->strides[j] < 1) return fail(AAE_ERR_INVALID, "stride %d of layer %d", d->strides[j], j);
            prod *= d->strides[j];
        }
        dh[i] = (int)(d->out_h / prod);
        dw[i] = (int)(d->out_w / prod);
        if (dh[i] < 1 || dw[i] < 1) return fail(AAE_ERR_INVALID, "layer %d: spatial size %dx%d", i, dh[i], dw[i]);
        if (d->num_filters[i] < 1) return fail(AAE_ERR_INVALID, "layer %d: %d filters", i, d->num_filters[i]);
    }
    aae_decoder* dec = new aae_decoder();
    dec->desc = *d;
    dec->h0 = dh[0]; dec->w0 = dw[0];
    aae_encoder* ctx = &dec->ctx;
    auto bail = [&](int rc) { aae_decoder_destroy(dec); return rc; };
    const float eps = d->bn_eps > 0.f ? d->bn_eps : 1e-3f;
    int wi = 0;

    Layer& D = dec->dense;
    D.H = D.W = D.Ho = D.Wo = 1; D.KS = 1; D.S = 1; D.pt = D.pl = 0; D.relu = 1;
    D.Cin = d->latent_size;
    D.Cout = dh[0] * dw[0] * d->num_filters[0];
    D.CoutPad = (int)align_up((size_t)D.Cout, 128);
    {
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(ctx, b, D.Cout, &D.bias)) return bail(rc);
        if (bn) if (int rc = upload_bn(ctx, hw, wi, D.Cout, eps, D)) return bail(rc);
        if (D.Cin % 32 == 0) {
            D.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, 1, D.Cin, D.Cout, D.CoutPad);
            if (int rc = upload(ctx, packed.data(), packed.size(), &D.wp)) return bail(rc);
        } else {
            D.kind = KIND_GENERIC;
            if (int rc = upload(ctx, k, (size_t)D.K() * D.Cout, &D.w_hwio)) return bail(rc);
        }
    }

    int H = dh[0], W = dw[0], C = d->num_filters[0];
    for (int i = 1; i <= L; ++i) {
        const bool last = i == L;
        DecStage s;
        Layer& Ls = s.L;
        s.UH = last ? d->out_h : dh[i];
        s.UW = last ? d->out_w : dw[i];
        Ls.H = H; Ls.W = W; Ls.Cin = C; Ls.Cout = last ? d->out_c : d->num_filters[i];
        Ls.KS = d->kernel_size; Ls.S = 1; s.pad = (Ls.KS - 1) / 2; Ls.pt = Ls.pl = s.pad;
        Ls.Ho = s.UH; Ls.Wo = s.UW;
        Ls.CoutPad = (int)align_up((size_t)Ls.Cout, 128);
        s.act = last ? aae::ACT_SIGMOID : aae::ACT_RELU;
        Ls.relu = last ? 0 : 1;
        s.nominal_flops_per_image = 2.0 * s.UH * s.UW * (double)Ls.KS * Ls.KS * Ls.Cin * Ls.Cout;
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(ctx, b, Ls.Cout, &Ls.bias)) return bail(rc);
        if (!last && bn) if (int rc = upload_bn(ctx, hw, wi, Ls.Cout, eps, Ls)) return bail(rc);

        const bool x2 = s.UH == 2 * H && s.UW == 2 * W;
        const bool x1 = s.UH == H && s.UW == W;
        const bool mfma_ok = Ls.Cin % 32 == 0;
        if (x2 && mfma_ok && !last && Ls.Cout > 4) {
            s.mode = STAGE_IGEMM_PHASED;
            const int d0 = phase_src_offset(0, 0, s.pad), d1 = phase_src_offset(1, 0, s.pad);
            s.U = phase_src_offset(0, Ls.KS - 1, s.pad) - d0 + 1;
            if (phase_src_offset(1, Ls.KS - 1, s.pad) - d1 + 1 != s.U) return bail(fail(AAE_ERR_RUNTIME, "phase tap counts differ"));
            s.ph_pt[0] = -d0; s.ph_pt[1] = -d1;
            std::vector<float> all;
            for (int ph = 0; ph < 4; ++ph) {
                const std::vector<float> wph = fold_phase(k, Ls.KS, s.pad, Ls.Cin, Ls.Cout, ph >> 1, ph & 1, s.U);
                const std::vector<float> packed = pack_weights(wph.data(), s.U * s.U, Ls.Cin, Ls.Cout, Ls.CoutPad);
                s.wp_phase_floats = (long long)packed.size();
                all.insert(all.end(), packed.begin(), packed.end());
            }
            if (int rc = upload(ctx, all.data(), all.size(), &s.wp_phases)) return bail(rc);
        } else if (x2 && mfma_ok && Ls.Cout <= 4 && !(bn && !last)) {
            s.mode = STAGE_NARROW_PHASED;
            int dlo = 0, dhi = 0, plo[2], phi[2];
            for (int p = 0; p < 2; ++p) {
                plo[p] = phase_src_offset(p, 0, s.pad);
                phi[p] = phase_src_offset(p, Ls.KS - 1, s.pad);
            }
            dlo = std::min(plo[0], plo[1]); dhi = std::max(phi[0], phi[1]);
            s.Un = dhi - dlo + 1; s.dmin = dlo;
            std::vector<float> wq((size_t)s.Un * s.Un * Ls.Cin * 16, 0.f);
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                const int Uy = phi[py] - plo[py] + 1, Ux = phi[px] - plo[px] + 1;
                if (Uy != Ux) return bail(fail(AAE_ERR_RUNTIME, "phase tap counts differ"));
                const std::vector<float> wph = fold_phase(k, Ls.KS, s.pad, Ls.Cin, Ls.Cout, py, px, Uy);
                for (int ty = 0; ty < Uy; ++ty)
                    for (int tx = 0; tx < Ux; ++tx) {
                        const int gy = ty + plo[py] - dlo, gx = tx + plo[px] - dlo;
                        for (int ci = 0; ci < Ls.Cin; ++ci)
                            for (int c = 0; c < Ls.Cout; ++c)
                                wq[(((size_t)(gy * s.Un + gx) * Ls.Cin + ci) * 4 + ph) * 4 + c] =
                                    wph[((size_t)(ty * Ux + tx) * Ls.Cin + ci) * Ls.Cout + c];
                    }
            }
            if (int rc = upload(ctx, wq.data(), wq.size(), &s.wq)) return bail(rc);
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < Ls.Cout; ++c) b4[c] = b[c];
            if (int rc = upload(ctx, b4, 4, &s.bias4)) return bail(rc);
        } else if (x1 && mfma_ok && !last && Ls.Cout > 4) {
            s.mode = STAGE_IGEMM_PLAIN;
            Ls.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, Ls.KS * Ls.KS, Ls.Cin, Ls.Cout, Ls.CoutPad);
            if (int rc = upload(ctx, packed.data(), packed.size(), &Ls.wp)) return bail(rc);
        } else {
            s.mode = STAGE_DIRECT;
            if (int rc = upload(ctx, k, (size_t)Ls.K() * Ls.Cout, &Ls.w_hwio)) return bail(rc);
        }
        dec->stages.push_back(s);
        H = s.UH; W = s.UW; C = Ls.Cout;
    }
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    *out = dec;
    return AAE_OK;
}

void aae_decoder_destroy(aae_decoder* dec) {
    if (!dec) return;
    for (void* p : dec->ctx.allocations) (void)hipFree(p);
    delete dec;
}

size_t aae_decoder_workspace_bytes(const aae_decoder* dec, int B) {
    if (!dec || B < 1) return 0;
    return aae_host::plan_decoder_workspace(dec, B).total;
}

int aae_decoder_forward(aae_decoder* dec, const float* z, int B, float* x_out, void* workspace, size_t ws_bytes, void* stream) {
    aae_host::Timer tm;
    return aae_host::decoder_forward_impl(dec, z, B, x_out, workspace, ws_bytes, stream, tm);
}

int aae_decoder_forward_timed(aae_decoder* dec, const float* z, int B, float* x_out, void* workspace, size_t ws_bytes,
                              void* stream, float* kernel_ms, int max_kernels, int* n_kernels) {
    using namespace aae_host;
    if (!kernel_ms || !n_kernels) return fail(AAE_ERR_INVALID, "aae_decoder_forward_timed: null output");
    Timer tm;
    tm.on = true;
    int rc = decoder_forward_impl(dec, z, B, x_out, workspace, ws_bytes, stream, tm);
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

const char* aae_decoder_kernel_label(const aae_decoder* dec, int i) {
    if (!dec || i < 0 || i >= (int)dec->ctx.records.size()) return "";
    return dec->ctx.records[i].label.c_str();
}

double aae_decoder_kernel_flops(const aae_decoder* dec, int i) {
    if (!dec || i < 0 || i >= (int)dec->ctx.records.size()) return 0.0;
    return dec->ctx.records[i].flops;
}

int aae_decoder_activation_info(const aae_decoder* dec, int B, int stage, size_t* offset_bytes, size_t* count) {
    using namespace aae_host;
    if (!dec || !offset_bytes || !count || B < 1) return fail(AAE_ERR_INVALID, "aae_decoder_activation_info: bad argument");
    const DecWorkspace ws = plan_decoder_workspace(dec, B);
    if (stage < 0 || stage >= (int)ws.act_off.size()) return fail(AAE_ERR_INVALID, "stage %d out of range", stage);
    *offset_bytes = ws.act_off[stage];
    *count = ws.act_count[stage];
    return AAE_OK;
}

}  // extern "C"
