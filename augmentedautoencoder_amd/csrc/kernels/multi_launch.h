// One launch per layer ACROSS objects: the problem table of the grouped multi-object query.
//
// The reference keeps one Augmented Autoencoder per object class in one process -- 30 for T-LESS
// (/root/reference/auto_pose/cfg_m3vision/m3_config_tless.cfg:10-39, m3_interface/ae_pose_estimator.py:61-78) -- and a
// frame's detections spread over many of them (:143-170: one session.run per box).  Answered object by object, a frame with C
// classes costs 6 C launches of the per-detection chain, each of which fills a fraction of the chip for 7-20 us and pays its
// own launch ramp, pipeline fill and tail.  Here every kernel of the chain takes a TABLE of problems -- per object its
// weights, activations, ticket words and tile counts -- in its kernel-argument segment, and the blocks of the grid are dealt
// out to the objects by a prefix sum: a frame costs six launches whatever C is, every launch holds C times the blocks, and
// the HBM streams of the C weight sets / codebooks run behind ONE ramp.
//
// Every block runs exactly the code of the per-object launch on exactly that launch's arguments (conv_first_block,
// conv_wavek_block, dense_gemv_block, scan_stream_block with the block index and block count of ITS object): same tiles,
// same K ranges, same summation orders, tickets keyed per (object, tile) because every object brings its own ticket words
// -> bit-identical to the per-object calls.  The table sits in the kernel arguments (scalar loads with a wave-uniform
// offset), not in device memory: nothing to upload, nothing to keep alive, capturable into a HIP graph.
#pragma once

namespace aae {

constexpr int kMultiMax = 16;              // objects per launch (the argument segment holds 4 KB); larger frames take several launches per layer

// blocks [first[o], first[o + 1]) of grid.x belong to object o (ranges may be padded: a block at or beyond its object's
// real block count idles -- keeps every object's first block on XCD 0 for xcd_remap)
struct MultiRange {
    int n;
    int first[kMultiMax + 1];
};
__device__ __forceinline__ int multi_find(const MultiRange& r, const int blk) {
    int o = 0;
#pragma unroll
    for (int i = 1; i < kMultiMax; ++i)
        if (i < r.n && blk >= r.first[i]) o = i;
    return o;
}

// Ticket words of one object's later launches (conv layers that split K, the dense GEMV's column tiles, the scan): installed
// with the call's nonce by an extra block of the first kernel, long before the first arrival (conv_wavek_f32.h, TicketPrep).
constexpr int kMultiPrepRanges = 6;
struct MultiTicketPrep {
    unsigned long long* words[kMultiPrepRanges];
    int count[kMultiPrepRanges];
    int n;
};
__device__ __forceinline__ void multi_ticket_prep_install(const MultiTicketPrep& t, const unsigned nonce) {
    for (int e = 0; e < t.n; ++e)
        for (int i = threadIdx.x; i < t.count[e]; i += blockDim.x) t.words[e][i] = (unsigned long long)nonce << 32;
}

}  // namespace aae
