// The codebook stage on the host side (/root/reference/auto_pose/ae/codebook.py:27,50,64-71: l2_normalize, matmul against the
// codebook, arg-max / upright / top-n): which scan kernel serves a query (plan_scan), the workspace layout, run_scan.
// Part of aae_hip_impl.h.
#pragma once

namespace aae_host {

// --------------------------------------------------------------- codebook side
struct ScanPlan {
    int nblk, Bpad, Bstride, Jpad, NT;
    bool gemv, stream;
    bool resident_ok;              // query-resident streaming kernel eligible (top-1, no similarity output, stride 1 decided at run time)
    int res_tiles_per_block, res_blocks, res_rh;
    bool topk_fused;               // top-k (2..8) inside the query-resident kernel: no [B][N] similarity matrix
    int cand_chunks;               // candidate lists per query that topk_merge_kernel merges
    size_t ticket_off, q_off, qp_off, pval_off, pidx_off, cs_off, cand_off, prune_off, total;
};

// answers of a top-1 stream scan that finishes inside its own launch (scan_ticket_finish)
struct ScanTicketOut {
    int64_t* idx_out = nullptr;
    float* score_out = nullptr;
    int idx_scale = 1;
    unsigned nonce = 0;            // != 0: the ticket words were prepared with this nonce by an earlier kernel on the stream
};

// masked: an upright query (col_stride > 1) WITHOUT a prepared compacted copy (aae_codebook_prepare_upright) -- every row is
// scanned and the rows off the stride are masked out.  The product build runs that rare form on the tile-resident kernels at
// every batch size (the masked variants of the stream kernels: experiments build); with the copy an upright query is an ordinary
// stride-1 scan of N / col_stride rows.
static ScanPlan plan_scan(const aae_codebook* cb, int B, int topk, bool masked = false) {
    ScanPlan s;
#ifdef AAE_EXPERIMENTS
    masked = false;
#endif
    s.nblk = ceil_div(cb->N, 128);
    s.Jpad = 128;
    s.stream = !masked && B <= 4 && (cb->scan_mode == AAE_SCAN_STREAM || cb->scan_mode == AAE_SCAN_AUTO);
    s.gemv = B <= 4 && cb->scan_mode == AAE_SCAN_GEMV;
    s.NT = B <= 32 ? 1 : (B <= 64 ? 2 : 4);
    s.Bpad = (int)align_up((size_t)B, (size_t)(32 * s.NT));
    if (cb->dtype == AAE_DTYPE_BF16) {           // B <= 4: HBM-streaming kernel (256 rows per block); else 64 queries per MFMA pass
        s.gemv = false;
        s.stream = !masked && B <= 4 && cb->scan_mode != AAE_SCAN_MFMA;
        s.Bpad = (int)align_up((size_t)B, (size_t)aae::kScanBf16QC);
        if (s.stream) s.nblk = ceil_div(cb->N, 256);
    }
    s.Bstride = s.Bpad;
    // B > 4: queries resident in registers, codebook streamed (codebook_scan_resident.h); about one block (8 waves)
    // per CU: row ranges x 128-query chunks.  Measured against the tile-resident kernels (whole nn call): B=8 0.035 ->
    // 0.024 ms, B=32 0.036 -> 0.024, B=256 0.106 -> 0.063; bf16 4x codebook B=32 0.083 -> 0.034, B=256 0.25 -> 0.078
    s.resident_ok = false; s.res_tiles_per_block = 0; s.res_blocks = 0; s.res_rh = 2;
    if (cb->scan_mode == AAE_SCAN_AUTO && cb->J == 128 && !s.stream && !s.gemv && B > 4) {
        // B > 128: 256 queries per block, every wave all rows of a tile (the codebook streamed once per 256 queries);
        // B <= 32, arg-max: FOUR waves share the rows of a tile for the one query group (with two, two of the CU's four matrix pipes sat
        // idle: 20.7 us per query of the 47 MB default codebook at any B <= 32, now 15.8; with two query groups -- 33 ... 64 queries -- all
        // eight waves are busy either way and the 128-row fp32 tiles in two LDS images measured slower, 23.3 against 21.3)
        s.res_rh = s.Bpad > 128 ? 1 : ((B <= 32 && topk == 1 && cb->scan_rh4) ? 4 : 2);
        const int tile_rows = (cb->dtype == AAE_DTYPE_BF16 || s.res_rh == 4) ? 128 : 64;
        const int ntiles = ceil_div(cb->N, tile_rows);
        const int qchunks = ceil_div(s.Bpad, 256 / s.res_rh);
        int row_blocks = (cb->cu_count > 0 ? cb->cu_count : 256) / qchunks;
        if (row_blocks < 1) row_blocks = 1;
        s.res_tiles_per_block = ceil_div(ntiles, row_blocks);
        if (s.res_tiles_per_block < 128 / tile_rows) s.res_tiles_per_block = 128 / tile_rows;   // never more row blocks than nblk
        s.res_blocks = ceil_div(ntiles, s.res_tiles_per_block);
        s.resident_ok = s.res_blocks <= s.nblk;          // the partial buffers are sized for nblk row blocks
    }
    // top-k (2 <= k <= 8) on the query-resident kernel: per-lane sorted lists instead of the [B][N] similarity matrix
    s.topk_fused = topk >= 2 && topk <= 8 && s.resident_ok;     // (AAE_SCAN_MFMA keeps the similarity-matrix path for A/B)
    size_t off = 0;
    s.ticket_off = off; off += align_up((size_t)aae::kTicketSlotWords * 8, 256);   // block_ticket_arrive words of the single-launch stream scan
    s.q_off = off;    off += align_up((size_t)B * cb->J * sizeof(float), 256);
    s.qp_off = off;   off += align_up((size_t)s.Jpad * s.Bpad * 6, 256);   // fp32 packing: 4 B/elem; bf16: 3 terms x 2 B
    // block partials: one row per scan block -- or per block of the persistent per-detection launch, whose grid (one block per
    // CU, detect_chain.h) can exceed the block count of a small codebook
    const int partial_rows = s.stream ? std::max(s.nblk, kChainMaxBlocks) : s.nblk;
    s.pval_off = off; off += align_up((size_t)partial_rows * s.Bstride * sizeof(float), 256);
    s.pidx_off = off; off += align_up((size_t)partial_rows * s.Bstride * sizeof(int), 256);
    s.cs_off = off;
    if (topk > 1 && !s.topk_fused) off += align_up((size_t)B * cb->N * sizeof(float), 256);
    s.cand_off = off;
    s.cand_chunks = s.topk_fused ? s.res_blocks : ceil_div(cb->N, aae::kTopKChunk);
    if (topk > 1) off += 2 * align_up((size_t)B * s.cand_chunks * topk * sizeof(float), 256);
    s.prune_off = off;                                  // shared bound words of the pruned top-k scan
    if (s.topk_fused) off += align_up((size_t)aae::kPruneReplicas * s.Bpad * aae::kPruneGroups * sizeof(int), 256);
    s.total = off;
    return s;
}

template <int NT>
static void launch_scan_mfma_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    constexpr int smem = aae::scan_mfma_smem<NT>();
    if (upright) {
        (void)hipFuncSetAttribute((const void*)aae::scan_mfma_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::scan_mfma_kernel<NT, true>), dim3(nblk), dim3(256), smem, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)aae::scan_mfma_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::scan_mfma_kernel<NT, false>), dim3(nblk), dim3(256), smem, stream, a);
    }
}

// (the similarity output is a template parameter of the stream kernels: as a run-time branch inside the row loop it split the
// loop into 64 basic blocks and serialised the cross-lane reductions)
template <int NQ>
static void launch_scan_stream_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    const int smem = NQ * 128 * (int)sizeof(float) + aae::kScanTicketSmem;
#ifdef AAE_EXPERIMENTS
    if (upright) {
        if (a.cs) AAE_LAUNCH((aae::scan_stream_kernel<NQ, true, true>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_kernel<NQ, true, false>), dim3(nblk), dim3(256), smem, stream, a);
        return;
    }
#endif
    (void)upright;                              // (product build: plan_scan never sends a masked query here)
    if (a.cs) AAE_LAUNCH((aae::scan_stream_kernel<NQ, false, true>), dim3(nblk), dim3(256), smem, stream, a);
    else AAE_LAUNCH((aae::scan_stream_kernel<NQ, false, false>), dim3(nblk), dim3(256), smem, stream, a);
}
#ifdef AAE_EXPERIMENTS
template <int NQ>
static void launch_scan_walk_t(const aae::ScanArgs& a, bool upright, int blocks, hipStream_t stream) {
    const int smem = 8 * NQ * (int)sizeof(float) + aae::kScanTicketSmem;
    if (a.cs) {
        if (upright) AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, true, true>), dim3(blocks), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, false, true>), dim3(blocks), dim3(256), smem, stream, a);
    } else {
        if (upright) AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, true, false>), dim3(blocks), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, false, false>), dim3(blocks), dim3(256), smem, stream, a);
    }
}
#endif
template <int NQ>
static void launch_scan_stream_bf16_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    const int smem = NQ * 256 * (int)sizeof(float) + aae::kScanTicketSmem;
#ifdef AAE_EXPERIMENTS
    if (upright) {
        if (a.cs) AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, true, true>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, true, false>), dim3(nblk), dim3(256), smem, stream, a);
        return;
    }
#endif
    (void)upright;
    if (a.cs) AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, false, true>), dim3(nblk), dim3(256), smem, stream, a);
    else AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, false, false>), dim3(nblk), dim3(256), smem, stream, a);
}

template <bool BF16, int K, int RH, bool NORM = false>
static void launch_scan_resident_t(const aae::ScanResidentArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int smem = aae::scan_resident_smem<BF16, RH>();
    (void)hipFuncSetAttribute((const void*)aae::scan_resident_kernel<BF16, K, RH, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    AAE_LAUNCH((aae::scan_resident_kernel<BF16, K, RH, NORM>), grid, dim3(aae::kScanResidentThreads), smem, stream, a);
}
template <bool BF16, int RH>
static void launch_scan_resident_k(const aae::ScanResidentArgs& a, dim3 grid, hipStream_t stream) {
    if (a.k <= 1 && a.z) launch_scan_resident_t<BF16, 0, RH, true>(a, grid, stream);      // the block normalises its own queries
    else if (a.k <= 1) launch_scan_resident_t<BF16, 0, RH>(a, grid, stream);
    else if (a.k <= 2) launch_scan_resident_t<BF16, 2, RH>(a, grid, stream);    // list slots: the smallest instantiated K >= k
    else if (a.k <= 4) launch_scan_resident_t<BF16, 4, RH>(a, grid, stream);
    else if (a.k == 5) launch_scan_resident_t<BF16, 5, RH>(a, grid, stream);
    else launch_scan_resident_t<BF16, 8, RH>(a, grid, stream);
}

// topk == 1: block partials (pval, pidx) for argmax_reduce_kernel; topk 2..8: candidate lists for topk_merge_kernel
static aae::ScanResidentArgs scan_resident_args(const aae_codebook* cb, const void* qp, int B, const ScanPlan& s, unsigned char* base, int topk, const float* raw_z,
                                                const ScanTicketOut* fin, dim3* grid_out) {
    aae::ScanResidentArgs a;
    a.E = cb->E; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4));
    a.qp = qp;
    a.z = raw_z;
    a.pval = reinterpret_cast<float*>(base + s.pval_off);
    a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
    a.N = cb->N; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride; a.tiles_per_block = s.res_tiles_per_block;
    const dim3 grid(s.res_blocks, ceil_div(s.Bpad, 256 / s.res_rh));
    a.k = topk > 1 ? topk : 0;
    if (fin && topk == 1 && grid.y == 1) {           // the last row block to arrive answers (no argmax_reduce launch)
        a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
        a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
    }
    if (topk > 1) {
        a.cand_v = reinterpret_cast<float*>(base + s.cand_off);
        a.cand_i = reinterpret_cast<int*>(base + s.cand_off + align_up((size_t)B * s.cand_chunks * topk * sizeof(float), 256));
        if (cb->topk_prune) a.prune = reinterpret_cast<int*>(base + s.prune_off);      // (reset by the normalise kernel in front)
    }
    *grid_out = grid;
    return a;
}
static int launch_scan_resident(const aae_codebook* cb, const void* qp, int B, const ScanPlan& s, unsigned char* base, hipStream_t stream,
                                int topk = 1, const float* raw_z = nullptr, const ScanTicketOut* fin = nullptr) {
    dim3 grid;
    const aae::ScanResidentArgs a = scan_resident_args(cb, qp, B, s, base, topk, raw_z, fin, &grid);
    const bool bf16 = cb->dtype == AAE_DTYPE_BF16;
    if (s.res_rh == 4) {                           // (arg-max only: plan_scan)
        if (bf16 && a.z) launch_scan_resident_t<true, 0, 4, true>(a, grid, stream);
        else if (bf16) launch_scan_resident_t<true, 0, 4>(a, grid, stream);
        else if (a.z) launch_scan_resident_t<false, 0, 4, true>(a, grid, stream);
        else launch_scan_resident_t<false, 0, 4>(a, grid, stream);
    } else if (bf16 && s.res_rh == 1) launch_scan_resident_k<true, 1>(a, grid, stream);
    else if (bf16) launch_scan_resident_k<true, 2>(a, grid, stream);
    else if (s.res_rh == 1) launch_scan_resident_k<false, 1>(a, grid, stream);
    else launch_scan_resident_k<false, 2>(a, grid, stream);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

// *partial_rows: how many [Bstride]-rows of (pval, pidx) the arg-max reduce has to look at
static int run_scan(aae_codebook* cb, const float* z, int B, int col_stride, float* cs_out, const ScanPlan& s,
                    unsigned char* base, hipStream_t stream, int* partial_rows = nullptr, const ScanTicketOut* fin = nullptr, int topk = 1) {
    float* q = reinterpret_cast<float*>(base + s.q_off);
    float* qp = reinterpret_cast<float*>(base + s.qp_off);
    const bool resident = s.resident_ok && cs_out == nullptr && col_stride == 1;
    if (partial_rows) *partial_rows = resident ? s.res_blocks : s.nblk;
    // arg-max on the query-resident kernel: the scan normalises the queries itself (no l2norm_pack launch in front)
    const ScanTicketOut* rfin = (resident && topk == 1 && !s.stream) ? fin : nullptr;          // (nn_impl passes fin for these only when the mode asks and the grid is one column of row blocks)
    if (resident && topk == 1 && cb->scan_fused_norm && ((uintptr_t)z & 15) == 0) return launch_scan_resident(cb, nullptr, B, s, base, stream, 1, z, rfin);
    if (cb->dtype == AAE_DTYPE_BF16 && s.stream) {
        aae::ScanArgs a;
        a.z = z; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * 2);
        a.E = cb->E; a.q = nullptr; a.qp = nullptr;
        a.pval = reinterpret_cast<float*>(base + s.pval_off);
        a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        a.cs = cs_out;
        a.N = cb->N; a.J = cb->J; a.Jpad = s.Jpad; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride;
        a.col_stride = col_stride;
        if (fin) {
            a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
            a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
        }
        const bool up = col_stride > 1;
        if (B == 1) launch_scan_stream_bf16_t<1>(a, up, s.nblk, stream);
        else if (B == 2) launch_scan_stream_bf16_t<2>(a, up, s.nblk, stream);
        else launch_scan_stream_bf16_t<4>(a, up, s.nblk, stream);
        AAE_HIP_TRY(hipGetLastError());
        return AAE_OK;
    }
    if (cb->dtype == AAE_DTYPE_BF16) {
        aae::L2NormBf16Args n;
        n.z = z; n.qp3 = reinterpret_cast<unsigned short*>(qp); n.B = B; n.J = cb->J; n.Jpad = 128; n.Bpad = s.Bpad;
        if (resident && topk > 1 && cb->topk_prune) n.prune = reinterpret_cast<int*>(base + s.prune_off);
        AAE_LAUNCH((aae::l2norm_pack_bf16x3_kernel), dim3(ceil_div(s.Bpad, 4)), dim3(256), 0, stream, n);
        AAE_HIP_TRY(hipGetLastError());
        if (resident) return launch_scan_resident(cb, n.qp3, B, s, base, stream, topk, nullptr, rfin);
        aae::ScanBf16Args a;
        a.E = reinterpret_cast<const unsigned short*>(cb->E);
        a.e_bytes = (unsigned)((size_t)cb->N * cb->J * 2);
        a.qp3 = n.qp3;
        a.pval = reinterpret_cast<float*>(base + s.pval_off);
        a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        a.cs = cs_out;
        a.N = cb->N; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride; a.col_stride = col_stride;
        if (col_stride > 1) {
            (void)hipFuncSetAttribute((const void*)aae::scan_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kScanBf16Smem);
            AAE_LAUNCH((aae::scan_bf16_kernel<true>), dim3(s.nblk), dim3(256), aae::kScanBf16Smem, stream, a);
        } else {
            (void)hipFuncSetAttribute((const void*)aae::scan_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kScanBf16Smem);
            AAE_LAUNCH((aae::scan_bf16_kernel<false>), dim3(s.nblk), dim3(256), aae::kScanBf16Smem, stream, a);
        }
        AAE_HIP_TRY(hipGetLastError());
        return AAE_OK;
    }
    if (!s.stream) {                     // the stream kernel normalises the queries itself
        aae::L2NormArgs n;
        n.z = z; n.q = q; n.qp = s.gemv ? nullptr : qp; n.B = B; n.J = cb->J; n.Jpad = s.Jpad; n.Bpad = s.gemv ? B : s.Bpad;
        if (resident && topk > 1 && cb->topk_prune) n.prune = reinterpret_cast<int*>(base + s.prune_off);
        AAE_LAUNCH((aae::l2norm_pack_kernel), dim3(ceil_div(n.Bpad, 4)), dim3(256), 0, stream, n);
        AAE_HIP_TRY(hipGetLastError());
    }
    if (resident) return launch_scan_resident(cb, qp, B, s, base, stream, topk, nullptr, rfin);

    aae::ScanArgs a;
    a.z = z; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * sizeof(float));
    a.E = cb->E; a.q = q; a.qp = qp;
    a.pval = reinterpret_cast<float*>(base + s.pval_off);
    a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
    a.cs = cs_out;
    a.N = cb->N; a.J = cb->J; a.Jpad = s.Jpad; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride;
    a.col_stride = col_stride;
    if (fin && s.stream) {
        a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
        a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
    }
    const bool upright = col_stride > 1;
#ifdef AAE_EXPERIMENTS
    if (s.stream && cb->scan_walk) {
        // one block per CU, never more blocks than 128-row groups (the partial buffers are sized for those)
        const int blocks = std::min(cb->cu_count > 0 ? cb->cu_count : 256, s.nblk);
        if (partial_rows) *partial_rows = blocks;
        if (B == 1) launch_scan_walk_t<1>(a, upright, blocks, stream);
        else if (B == 2) launch_scan_walk_t<2>(a, upright, blocks, stream);
        else if (B == 3) launch_scan_walk_t<3>(a, upright, blocks, stream);
        else launch_scan_walk_t<4>(a, upright, blocks, stream);
    } else
#endif
    if (s.stream) {
        if (B == 1) launch_scan_stream_t<1>(a, upright, s.nblk, stream);
        else if (B == 2) launch_scan_stream_t<2>(a, upright, s.nblk, stream);
        else if (B == 3) launch_scan_stream_t<3>(a, upright, s.nblk, stream);
        else launch_scan_stream_t<4>(a, upright, s.nblk, stream);
    }
#ifdef AAE_EXPERIMENTS
    else if (s.gemv) {
        const int smem = 2 * 4 * 4 * (int)sizeof(float);
        if (upright) AAE_LAUNCH((aae::scan_gemv_kernel<4, true>), dim3(s.nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_gemv_kernel<4, false>), dim3(s.nblk), dim3(256), smem, stream, a);
    }
#endif
    else if (s.NT == 1) launch_scan_mfma_t<1>(a, upright, s.nblk, stream);
    else if (s.NT == 2) launch_scan_mfma_t<2>(a, upright, s.nblk, stream);
    else launch_scan_mfma_t<4>(a, upright, s.nblk, stream);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

}  // namespace aae_host
