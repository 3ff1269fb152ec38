// Host side of the grouped multi-object query (include/aae_hip.h: aae_encode_nn_multi, aae_detect_nn_multi,
// aae_codebook_nn_multi): a frame's detections of SEVERAL object classes -- each class its own encoder weights and its own
// codebook, as /root/reference/auto_pose/m3_interface/ae_pose_estimator.py:61-78 keeps them -- answered with one launch per
// LAYER instead of one six-launch chain per class (:143-170 runs one session.run per box).  Included at the end of
// aae_hip_impl.h; kernels: kernels/multi_launch.h and the *_multi_kernel forms beside the per-object kernels.
//
// Which items are grouped.  An item (object, n detections) joins a group when its per-object call would run the
// per-detection chain -- conv1 in the one-block-per-pixel-group form, every later conv layer on the wave-split-K kernel
// (4 waves, 2 slabs in flight), the dense layer as the ticketed GEMV, the fp32 stream scan with its in-launch finish --
// i.e. n <= 4 of the reference network at the default options.  Items of one group share the batch n and the network shape
// (so that one kernel instantiation serves all of them); each keeps ITS launch plan -- tiles, K splits, tail cut -- and gets
// its own slice of the workspace (activations, partials, ticket words), so the answers are the per-object calls' bit for bit.
// Mid batches (5 or more detections per object; the reference's one-AAE-per-class frame at B = 256 over 8 objects holds ~32 each,
// SURVEY section 8d config 4): objects of one network shape whose conv layers all run as polyphase Winograd share ONE launch per conv layer
// (conv_wino_layer_multi_kernel: the block -> object map is a prefix sum over the objects' window regions) where the GROUP's blocks
// fill the rounds of blocks they occupy -- eight buckets of 32 crops fill the chip like one batch of 256, where each bucket alone leaves
// conv4 (four images per block: 64 blocks) a quarter of a round.  conv1, the dense layer and the scan run per object around them.  Every
// block runs exactly its object's single-launch block: each layer bit-identical to the object's own Winograd launch -- an object whose
// own call would take the direct kernels for a layer (too few blocks alone) differs from that call by the two forms' fp32 rounding
// (<= 2.3e-6 of the latent scale, conv_winograd_f32.h).
// Everything else (bf16 codebooks, split precision, other networks or options) is answered by the per-object path inside the same call,
// one item after the other in a shared slice.
#pragma once

namespace aae_host {

struct MultiItemPlan {
    bool grouped = false;
    bool mid = false;                          // member of a mid-batch group (one Winograd launch per conv layer across the objects)
    int n = 0, row0 = 0;                       // detections of the item, its first row in the concatenated inputs / outputs
    size_t enc_off = 0, enc_bytes = 0, cb_off = 0, cb_bytes = 0;     // grouped items: own workspace slices
    Workspace ws;
    std::vector<WaveKPlan> plans;              // per conv layer (grouped items)
    std::vector<WaveKPlan> rem_plans;          // mid items, per conv layer: the wave-split-K plan of the last n mod 4 images where the group hands them over (use = false: none)
    size_t rem_partial_off = 0, rem_partial_bytes = 0;      // ... and their partial sums (behind the item's own workspace)
    ScanPlan sp;
    const aae_codebook* eff = nullptr;         // the codebook the scan runs over (the compacted upright copy when col_stride > 1)
    int idx_scale = 1;
    std::vector<int> sig;                      // items with equal signatures share kernel instantiations
};

struct MultiPlan {
    std::vector<MultiItemPlan> items;
    std::vector<std::vector<int>> groups;      // grouped items that share launches: equal signatures, at most kMultiMax members, item order
    std::vector<std::vector<int>> mid_groups;  // mid-batch items that share their Winograd launches
    std::vector<std::vector<char>> group_wino; // [group][conv layer]: the layer runs as one Winograd launch across the group's objects (multi_group_winograd)
    std::vector<std::vector<char>> mid_wino;   // [mid group][conv layer]: the layer runs as one Winograd launch across the group (the others per object)
    std::vector<std::vector<char>> mid_rem;    // [mid group][conv layer]: the incomplete four-image blocks leave the Winograd launch (multi_mid_ragged)
    size_t seq_enc_off = 0, seq_enc_bytes = 0, seq_cb_off = 0, seq_cb_bytes = 0, total = 0;
    int rows = 0;
};

static thread_local int t_multi_launches = 0;      // kernel launches queued by the calling thread's last multi call (its grouped part)

static int wavek_shape_key(const WaveKPlan& w) { return (w.MT == 1 ? 1000 : 0) + w.NT * 100 + w.waves * 10 + w.depth; }

// Can the scan of (cb, n detections, col_stride) run as one object's share of scan_stream_multi_kernel?
static bool multi_scan_groupable(const aae_codebook* cb, int n, int col_stride, const aae_codebook** eff_out, int* idx_scale) {
    const aae_codebook* eff = cb;
    *idx_scale = 1;
    if (col_stride > 1) {
        if (!(cb->upright && cb->upright_stride == col_stride)) return false;      // (the masked full scan: per-object path)
        eff = cb->upright;
        *idx_scale = col_stride;
    }
    *eff_out = eff;
    if (n < 1 || n > 4 || eff->dtype != AAE_DTYPE_F32 || eff->scan_ticket == 0 || eff->scan_walk) return false;
    return plan_scan(eff, n, 1).stream;
}

static bool multi_encoder_groupable(const aae_encoder* enc, int n, std::vector<WaveKPlan>& plans, std::vector<int>& sig) {
    const size_t nl = enc->layers.size();
    if (n < 1 || n > 4 || nl < 2 || runs_split(enc, n) || enc->detect_chain || enc->wavek_ablate || enc->wavek_timeline) return false;
    const Layer& L0 = enc->layers[0];
    if (L0.kind != KIND_FIRST_MFMA || n * ceil_div(L0.Ho * L0.Wo, 128) > enc->first_group_split_max_tiles) return false;
    const Layer& D = enc->dense;
    if (D.kind != KIND_IGEMM || !enc->dense_gemv || n > gemv_max_batch(enc) || D.K() % aae::kGemvChunk != 0 || !gemv_uses_ticket(enc, D)) return false;
    plans.assign(nl, WaveKPlan());
    sig.clear();
    sig.push_back(enc->multi_group_plan ? 0 : n);   // per-object plans: one group per detection count (MQ / NQ of the GEMV and the scan are part of the
                                                   // per-object call's summation order); group plans: objects with 1 ... 4 detections share their launches
    sig.push_back(enc->multi_group_plan);      // ... and the options the group's plan is made from
    sig.push_back(enc->wavek_spread); sig.push_back(enc->wavek_g_boost); sig.push_back(wavek_round_blocks(enc)); sig.push_back(enc->wavek_eff64x32_pct);
    sig.push_back(enc->multi_group_winograd); sig.push_back(enc->winograd); sig.push_back(enc->winograd_min_fill_pct); sig.push_back(enc->winograd_min_blocks); sig.push_back(enc->winograd_xcd_cols);
    sig.push_back(enc->compact_workspace); sig.push_back(enc->first_vec4); sig.push_back(enc->multi_force_shape); sig.push_back(enc->multi_force_g); sig.push_back(enc->multi_xcd_affine); sig.push_back(enc->multi_force_depth);
    const int32_t* d = reinterpret_cast<const int32_t*>(&enc->desc);
    for (size_t i = 0; i < sizeof(aae_encoder_desc) / sizeof(int32_t); ++i) sig.push_back(d[i]);      // (bn_eps as its bit pattern)
    for (size_t li = 1; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        if (L.kind != KIND_IGEMM) return false;
        const WaveKPlan w = plan_wavek(enc, L, (long long)n * L.Ho * L.Wo, false);
        if (!w.use || w.waves != 4 || w.depth != 2) return false;
        const int key = wavek_shape_key(w);
        const bool spread = key == 1142 ? (enc->wavek_spread & 2) != 0 : (key == 242 ? (enc->wavek_spread & 1) != 0 : false);
        if (!((key == 1142 && spread) || key == 142 || (key == 242 && spread))) return false;         // (the instantiated grouped forms = the defaults)
        plans[li] = w;
        sig.push_back(enc->multi_group_plan ? 0 : key);
    }
    return true;
}

// The launch plan of one conv layer for a GROUP of `members` objects (option "multi_group_plan", default on).  The per-object plan
// cuts ONE object's layer into a chip-full of blocks -- at one detection that means the smallest wave tile (32 x 32: twice the
// operand bytes per MFMA of the 64 x 64 tile) and K splits that leave every wave a dozen slabs behind a four-slab pipeline fill.
// A group brings `members` times the tiles, so the wave tile and the K split are chosen for the group's TOTAL tile count, and
// every member runs that plan.  The rule is the outcome of a sweep of every (wave tile, K split) per layer on MI355X at 2 ... 16
// objects x 1 ... 4 detections (tools/multi_plan_ab.py -> profiles/r13/multi_plan_ab.jsonl; the planner-by-cost's estimate,
// fitted to single-object mid batches, mis-ranks these launches by up to 8 %):
//   * 64 x 64 wave tiles, K cut so that the group fills TWO blocks per compute unit (the kernel's launch bound; 0.70 of the
//     matrix peak measured, the best any wave-split-K form reaches) ...
//   * ... unless that needs K cut four or more ways (few, long-K tiles: conv4 of the reference network, 400 slabs): then
//     64 x 32 tiles at ONE block per compute unit -- half the K cuts for the same bytes per tile, a quarter of the partial
//     sums to hand over (8 objects x 1 detection: conv4 99.8 -> 68 us);
//   * the 32 x 32 tile never wins once two objects share a launch.
// Same fma chains per output element up to where K is cut: results differ from the per-object call's by fp32 summation
// order only (like a batch of another size does); "multi_group_plan" = 0 keeps every object on its own plan, bit-identical
// to aae_encode_nn.
// rows[k] = GEMM rows (detections x Ho x Wo) of member k's layer; the members share (wave tile, K cut), each has its own tile count
static std::vector<WaveKPlan> plan_wavek_group(const aae_encoder* enc, const Layer& L, const std::vector<long long>& rows) {
    const int slabs = (int)(L.K() / 32), waves = 4, cus = wavek_round_blocks(enc);
    const int layer = L.index > 0 ? L.index - 1 : 0;
    const int force = (enc->multi_force_shape >> (4 * layer)) & 15;          // (A/B: one nibble per conv layer, 1 = 32 x 32, 2 = 64 x 32, 3 = 64 x 64)
    const int force_g = (enc->multi_force_g >> (8 * layer)) & 255;           // (A/B: one byte per conv layer)
    const int force_depth = (enc->multi_force_depth >> (4 * layer)) & 15;    // (A/B: one nibble per conv layer: 2 | 3 slabs in flight, 64 x 32 tiles only)
    auto make = [&](int mt, int nt, int blocks_per_cu) {
        std::vector<WaveKPlan> out(rows.size());
        long long total = 0, most = 0;
        for (long long M : rows) {
            const long long tiles_o = ((M + 32 * mt - 1) / (32 * mt)) * (long long)(L.CoutPad / (32 * nt));
            total += tiles_o;
            most = std::max(most, tiles_o);
        }
        if (most > kWaveKTileCap || total > (1 << 20)) return std::vector<WaveKPlan>();
        int g = force_g ? force_g : (int)((long long)cus * blocks_per_cu / total);
        const int gmax = std::min(slabs / (2 * waves), (int)aae::kTicketSingleLevelMax);
        if (g > gmax) g = gmax;
        if (g < 1 || most > kLayerTicketWords) g = 1;
        for (size_t k = 0; k < rows.size(); ++k) {
            WaveKPlan& w = out[k];
            w.use = true;
            w.MT = mt; w.NT = nt; w.waves = waves;
#ifdef AAE_EXPERIMENTS
            w.depth = (mt == 2 && nt == 1 && force_depth == 3) ? 3 : 2;
#else
            w.depth = 2;
            (void)force_depth;
#endif
            w.num_mt = (int)((rows[k] + 32 * mt - 1) / (32 * mt));
            w.num_nt = L.CoutPad / (32 * nt);
            w.gsplits = g;
            w.tail_tiles = 0; w.tail_g = 1;
            w.partial_bytes = g > 1 ? (size_t)w.blocks() * (mt * nt * 16) * 64 * sizeof(float) : 0;
        }
        return out;
    };
    const bool have64 = (enc->wavek_spread & 1) != 0, have32 = (enc->wavek_spread & 2) != 0;      // (the instantiated grouped forms)
    if (force == 1 && have32) return make(1, 1, 2);
    if (force == 2) return make(2, 1, 1);
    if (force == 3 && have64) return make(2, 2, 2);
    std::vector<WaveKPlan> a = have64 ? make(2, 2, 2) : std::vector<WaveKPlan>();
    if (!a.empty() && a[0].gsplits < 4) return a;
    std::vector<WaveKPlan> b = make(2, 1, 1);
    return b.empty() ? a : b;
}

// workspace slice of a grouped item: ticket words, one activation buffer per conv layer, the largest partial buffer of its plans
static Workspace plan_workspace_grouped(const aae_encoder* enc, int n, const std::vector<WaveKPlan>& plans) {
    Workspace ws;
    size_t off = kTicketBytes;
    size_t partial = gemv_partial_bytes(enc->dense, n);
    for (size_t li = 0; li < enc->layers.size(); ++li) {
        const Layer& L = enc->layers[li];
        ws.act_off.push_back(off);
        off += align_up((size_t)n * L.Ho * L.Wo * L.Cout * sizeof(float), 256);
        if (li < plans.size() && plans[li].use) partial = std::max(partial, plans[li].partial_bytes);
    }
    ws.partial_off = off;
    ws.partial_bytes = partial;
    off += align_up(partial, 256);
    ws.total = off;
    return ws;
}

// Can (enc, n detections) join a mid-batch group?  Default options in exact fp32, every conv layer behind the first one prepared for Winograd.
// (weights_pending: asked before the Winograd-domain weights exist -- aae_multi_workspace_bytes builds them only for objects whose group would form)
static bool multi_encoder_mid_groupable(const aae_encoder* enc, int n, std::vector<int>& sig, bool weights_pending = false) {
    const size_t nl = enc->layers.size();
    if (n < 5 || nl < 2 || enc->winograd != 1 || enc->winograd_wide || !enc->multi_mid_group || runs_split(enc, n)) return false;
    for (size_t li = 1; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        if (L.kind != KIND_IGEMM || L.wino_geom < 0 || (!L.wino[0] && !weights_pending)) return false;      // (the weights: prepared by aae_multi_workspace_bytes)
        if ((unsigned long long)n * L.H * L.W * L.Cin * sizeof(float) >= 0x7FFFFF00ull) return false;
    }
    sig.clear();
    sig.push_back(enc->winograd_min_fill_pct); sig.push_back(enc->winograd_min_blocks); sig.push_back(enc->winograd_xcd_cols); sig.push_back(wavek_round_blocks(enc));
    sig.push_back(enc->multi_mid_ragged); sig.push_back(enc->wavek); sig.push_back(enc->wavek_spread); sig.push_back(enc->multi_force_shape); sig.push_back(enc->multi_force_g);
    const int32_t* d = reinterpret_cast<const int32_t*>(&enc->desc);
    for (size_t i = 0; i < sizeof(aae_encoder_desc) / sizeof(int32_t); ++i) sig.push_back(d[i]);
    return true;
}
static int wino_regions(const Layer& L, int n) { return L.wino_geom == 0 ? (L.Ho / 16) * (L.Wo / 16) * n : ceil_div(n, 4); }
// A layer of four-image blocks: do the objects' incomplete blocks (n mod 4 images) open one more round of blocks than the complete ones need?  Then they are worth a launch
// of their own on the direct kernel (plan_mid_ragged) and the Winograd launch holds the complete blocks only.
static bool mid_ragged_opens_a_round(const aae_encoder* enc0, const Layer& L, const std::vector<int>& counts, long long* complete_regions) {
    if (L.wino_geom != 1 || !enc0->multi_mid_ragged || !enc0->wavek) return false;
    long long with = 0, without = 0;
    for (int n : counts) { with += ceil_div(n, 4); without += n / 4; }
    const long long cus = wavek_round_blocks(enc0);
    const int nbn = L.Cout / 64, xc = wino_xcd_cols(enc0, L);
    const long long rounds_with = ((long long)aae::wino_grid_blocks((int)with, nbn, xc) + cus - 1) / cus;
    const long long rounds_without = ((long long)aae::wino_grid_blocks((int)without, nbn, xc) + cus - 1) / cus;
    if (complete_regions) *complete_regions = without;
    return with > without && without >= 1 && rounds_without < rounds_with;
}
// ... and which conv layers does the GROUP fill the chip on (the round-fill rule of runs_winograd, on the group's blocks)?  Those run as one Winograd launch across the objects;
// the others (4 classes x 6 boxes: conv4's 64 blocks) run per object on whatever kernel the object's own forward takes.  A group forms when at least one layer passes.
static std::vector<char> mid_group_layers(const aae_encoder* enc0, const std::vector<int>& counts) {
    const long long cus = wavek_round_blocks(enc0);
    std::vector<char> pass(enc0->layers.size(), 0);
    for (size_t li = 1; li < enc0->layers.size(); ++li) {
        const Layer& L = enc0->layers[li];
        long long regions = 0;
        for (int n : counts) regions += wino_regions(L, n);
        long long complete = 0;
        if (mid_ragged_opens_a_round(enc0, L, counts, &complete)) regions = complete;      // (then the rule looks at the complete blocks: the others leave the launch)
        const long long blocks = regions * (L.Cout / 64), rounds = (blocks + cus - 1) / cus;
        if (enc0->winograd_min_blocks > 0) pass[li] = blocks >= enc0->winograd_min_blocks;   // (tests, A/B: a plain block count instead of the fill rule, as in runs_winograd)
        else pass[li] = 100 * blocks >= (long long)enc0->winograd_min_fill_pct * rounds * cus;
    }
    return pass;
}
static bool mid_group_fills(const aae_encoder* enc0, const std::vector<int>& counts) {
    for (char c : mid_group_layers(enc0, counts))
        if (c) return true;
    return false;
}

// Layers of four-image blocks (8 x 8 outputs) in a mid-batch group: an object with n mod 4 != 0 ends in a block with empty image slots.  Where those blocks
// open one more ROUND of blocks than the complete ones need (config 4: {34, 26, 27, 32, 31, 32, 33, 41} -> 67 groups x 8 column blocks = 536 = three rounds for 24
// blocks; the 61 complete groups = 488 fit two), the last n mod 4 images of every object go to ONE grouped wave-split-K launch (the per-detection chain's kernel,
// a plan for the group: plan_wavek_group) behind the Winograd launch: 1.28 -> ~0.95 ms for conv4 of that frame.
static bool wavek_multi_instantiated(const aae_encoder* enc, const WaveKPlan& w) {
    const int key = wavek_shape_key(w);
    return w.use && w.waves == 4 && w.depth == 2 && ((key == 1142 && (enc->wavek_spread & 2)) || key == 142 || (key == 242 && (enc->wavek_spread & 1)));
}
static void plan_mid_ragged(const aae_multi_item* items, MultiPlan& mp) {
    mp.mid_rem.assign(mp.mid_groups.size(), std::vector<char>());
    mp.mid_wino.assign(mp.mid_groups.size(), std::vector<char>());
    for (size_t gi = 0; gi < mp.mid_groups.size(); ++gi) {
        const std::vector<int>& g = mp.mid_groups[gi];
        const aae_encoder* enc0 = items[g[0]].enc;
        const size_t nl = enc0->layers.size();
        mp.mid_rem[gi].assign(nl, 0);
        {
            std::vector<int> all;
            for (int i : g) all.push_back(mp.items[(size_t)i].n);
            mp.mid_wino[gi] = mid_group_layers(enc0, all);
        }
        for (int i : g) mp.items[(size_t)i].rem_plans.assign(nl, WaveKPlan());
        for (size_t li = 1; li < nl; ++li) {
            const Layer& L = enc0->layers[li];
            std::vector<long long> rows;
            std::vector<int> who, counts;
            for (int i : g) {
                const int n = mp.items[(size_t)i].n;
                counts.push_back(n);
                if (n % 4) { rows.push_back((long long)(n % 4) * L.Ho * L.Wo); who.push_back(i); }
            }
            if (!mp.mid_wino[gi][li] || !mid_ragged_opens_a_round(enc0, L, counts, nullptr)) continue;
            const std::vector<WaveKPlan> gp = plan_wavek_group(enc0, L, rows);
            if (gp.size() != who.size() || !wavek_multi_instantiated(enc0, gp[0])) continue;
            bool ok = true;
            for (const WaveKPlan& w : gp) ok = ok && wavek_shape_key(w) == wavek_shape_key(gp[0]) && w.num_mt * w.num_nt <= kLayerTicketWords;
            if (!ok) continue;
            mp.mid_rem[gi][li] = 1;
            for (size_t k = 0; k < who.size(); ++k) {
                MultiItemPlan& p = mp.items[(size_t)who[k]];
                p.rem_plans[li] = gp[k];
                p.rem_partial_bytes = std::max(p.rem_partial_bytes, gp[k].partial_bytes);
            }
        }
    }
}

// Layout of one call: [shared slice of the per-object path: encoder part, codebook part][grouped item 0: encoder, codebook][item 1] ...
static int plan_multi(const aae_multi_item* items, int n_items, bool scan_only, MultiPlan& mp, bool weights_pending = false) {
    if (!items || n_items < 1) return fail(AAE_ERR_INVALID, "multi-object query: no items");
    mp.items.assign((size_t)n_items, MultiItemPlan());
    int row = 0;
    for (int i = 0; i < n_items; ++i) {
        const aae_multi_item& it = items[i];
        MultiItemPlan& p = mp.items[(size_t)i];
        if (!it.cb || (!scan_only && !it.enc)) return fail(AAE_ERR_INVALID, "multi-object query: item %d has a null handle", i);
        if (it.n < 1) return fail(AAE_ERR_INVALID, "multi-object query: item %d has %d detections (want >= 1)", i, it.n);
        if (it.col_stride < 1) return fail(AAE_ERR_INVALID, "multi-object query: item %d col_stride %d < 1", i, it.col_stride);
        if (!scan_only && it.cb->J != it.enc->desc.latent_size)
            return fail(AAE_ERR_INVALID, "multi-object query: item %d pairs a %d-d encoder with a %d-d codebook", i, it.enc->desc.latent_size, it.cb->J);
        if (scan_only && it.cb->J != items[0].cb->J)
            return fail(AAE_ERR_INVALID, "multi-object query: item %d has latent size %d, item 0 has %d (one [rows, J] latent array serves all)", i, it.cb->J, items[0].cb->J);
        if (!scan_only && it.enc->desc.latent_size != items[0].enc->desc.latent_size)
            return fail(AAE_ERR_INVALID, "multi-object query: item %d has latent size %d, item 0 has %d (one [rows, J] latent array serves all)", i,
                        it.enc->desc.latent_size, items[0].enc->desc.latent_size);
        p.n = it.n;
        p.row0 = row;
        row += it.n;
        p.grouped = multi_scan_groupable(it.cb, it.n, it.col_stride, &p.eff, &p.idx_scale) &&
                    (scan_only || multi_encoder_groupable(it.enc, it.n, p.plans, p.sig));
        if (scan_only) p.sig.assign(1, it.n);
        if (p.grouped) {
            p.sp = plan_scan(p.eff, it.n, 1);
            p.cb_bytes = align_up(p.sp.total, 256);
        } else if (!scan_only && multi_encoder_mid_groupable(it.enc, it.n, p.sig)) {
            p.mid = true;                           // (a candidate: confirmed below once its group is known)
        }
    }
    // mid-batch groups: equal signatures, at most kMultiMax members, at least two, and the group must fill the chip on every conv layer
    mp.mid_groups.clear();
    for (size_t i = 0; i < mp.items.size(); ++i) {
        if (!mp.items[i].mid) continue;
        bool placed = false;
        for (auto& g : mp.mid_groups)
            if ((int)g.size() < aae::kMultiMax && mp.items[(size_t)g[0]].sig == mp.items[i].sig) { g.push_back((int)i); placed = true; break; }
        if (!placed) mp.mid_groups.push_back(std::vector<int>(1, (int)i));
    }
    for (size_t gi = 0; gi < mp.mid_groups.size();) {
        const std::vector<int>& g = mp.mid_groups[gi];
        std::vector<int> counts;
        for (int i : g) counts.push_back(mp.items[(size_t)i].n);
        if (g.size() >= 2 && mid_group_fills(items[g[0]].enc, counts)) { ++gi; continue; }
        for (int i : g) mp.items[(size_t)i].mid = false;
        mp.mid_groups.erase(mp.mid_groups.begin() + (long)gi);
    }
    plan_mid_ragged(items, mp);
    for (int i = 0; i < n_items; ++i) {
        const aae_multi_item& it = items[i];
        MultiItemPlan& p = mp.items[(size_t)i];
        if (p.grouped) continue;
        if (p.mid) {
            p.ws = plan_workspace(it.enc, it.n);
            p.rem_partial_off = align_up(p.ws.total, 256);
            p.enc_bytes = p.rem_partial_off + align_up(p.rem_partial_bytes, 256);
            p.cb_bytes = align_up(plan_scan(it.cb, it.n, 1).total, 256);
        } else {
            if (!scan_only) mp.seq_enc_bytes = std::max(mp.seq_enc_bytes, align_up(plan_workspace(it.enc, it.n).total, 256));
            mp.seq_cb_bytes = std::max(mp.seq_cb_bytes, align_up(plan_scan(it.cb, it.n, 1).total, 256));
        }
    }
    // groups: equal signatures, at most kMultiMax members each, in item order
    mp.groups.clear();
    for (size_t i = 0; i < mp.items.size(); ++i) {
        if (!mp.items[i].grouped) continue;
        bool placed = false;
        for (auto& g : mp.groups)
            if ((int)g.size() < aae::kMultiMax && mp.items[(size_t)g[0]].sig == mp.items[i].sig) { g.push_back((int)i); placed = true; break; }
        if (!placed) mp.groups.push_back(std::vector<int>(1, (int)i));
    }
    if (!scan_only)
        for (const std::vector<int>& g : mp.groups) {
            const aae_encoder* enc0 = items[g[0]].enc;
            if (g.size() >= 2 && enc0->multi_group_plan) {               // the group's own plan (every member the same network shape)
                for (size_t li = 1; li < enc0->layers.size(); ++li) {
                    const Layer& L = enc0->layers[li];
                    std::vector<long long> rows;
                    for (int i : g) rows.push_back((long long)mp.items[(size_t)i].n * L.Ho * L.Wo);
                    const std::vector<WaveKPlan> gp = plan_wavek_group(enc0, L, rows);
                    if (gp.size() == g.size())
                        for (size_t k = 0; k < g.size(); ++k) mp.items[(size_t)g[k]].plans[li] = gp[k];
                }
            }
            for (int i : g) {
                MultiItemPlan& p = mp.items[(size_t)i];
                p.ws = plan_workspace_grouped(items[i].enc, p.n, p.plans);
                p.enc_bytes = align_up(p.ws.total, 256);
            }
        }
    // per-detection groups: conv layers whose blocks -- over ALL objects of the group -- pass the fill rule run as one Winograd launch (a frame of 8 classes x 4 boxes: conv2 512 blocks,
    // conv3 256; 8 x 1: conv2 128 = half a round: stays on the wave-split-K kernel)
    mp.group_wino.assign(mp.groups.size(), std::vector<char>());
    if (!scan_only)
        for (size_t gi = 0; gi < mp.groups.size(); ++gi) {
            const std::vector<int>& g = mp.groups[gi];
            const aae_encoder* enc0 = items[g[0]].enc;
            const size_t nl = enc0->layers.size();
            mp.group_wino[gi].assign(nl, 0);
            if (g.size() < 2 || !enc0->multi_group_plan || !enc0->multi_group_winograd || enc0->winograd != 1 || enc0->winograd_wide) continue;
            const long long cus = wavek_round_blocks(enc0);
            for (size_t li = 1; li < nl; ++li) {
                const Layer& L0 = enc0->layers[li];
                if (L0.kind != KIND_IGEMM || L0.wino_geom < 0) continue;
                long long regions = 0, images = 0;
                bool ready = true;
                for (int i : g) {
                    const Layer& L = items[i].enc->layers[li];
                    ready = ready && L.wino_geom == L0.wino_geom && (L.wino[0] || weights_pending);
                    regions += wino_regions(L, mp.items[(size_t)i].n);
                    images += mp.items[(size_t)i].n;
                }
                if (!ready) continue;
                // (four-image blocks: an object with one box fills a quarter of its block -- the block count says nothing about the work then; with at most 16 objects
                //  per group such a layer never reaches the rule today, the guard keeps it that way)
                if (L0.wino_geom == 1 && enc0->winograd_min_blocks == 0 && 4 * images < 3 * 4 * regions) continue;
                const long long blocks = aae::wino_grid_blocks((int)regions, L0.Cout / 64, wino_xcd_cols(enc0, L0)), rounds = (blocks + cus - 1) / cus;
                const bool fills = enc0->winograd_min_blocks > 0 ? blocks >= enc0->winograd_min_blocks : 100 * blocks >= (long long)enc0->winograd_min_fill_pct * rounds * cus;
                if (fills) mp.group_wino[gi][li] = 1;
            }
        }
    mp.rows = row;
    size_t off = 0;
    mp.seq_enc_off = off; off += mp.seq_enc_bytes;
    mp.seq_cb_off = off; off += mp.seq_cb_bytes;
    for (MultiItemPlan& p : mp.items)
        if (p.grouped || p.mid) {
            p.enc_off = off; off += p.enc_bytes;
            p.cb_off = off; off += p.cb_bytes;
        }
    mp.total = off;
    return AAE_OK;
}

template <int KS, int C>
static void launch_first_multi_t(const aae::ConvFirstMultiArgs& m, bool u8, bool vec4, dim3 grid, int smem, hipStream_t stream) {
    // (the dynamic-LDS ceiling of every instantiation is raised ONCE per process, to the CU's 160 KB: not a runtime call per grouped frame)
    static const bool once = ((void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              (void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              (void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    if (u8 && vec4) AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, true, true>), grid, dim3(256), smem, stream, m);
    else if (u8) AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, true, false>), grid, dim3(256), smem, stream, m);
    else AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, false, false>), grid, dim3(256), smem, stream, m);
}

// whole-tile form of the grouped conv1 (mid-batch groups): no ticket preparation blocks, grid.z = 1
template <int KS, int C>
static void launch_first_multi_tiles_t(const aae::ConvFirstMultiArgs& m, bool u8, bool vec4, dim3 grid, int smem, hipStream_t stream) {
    static const bool once = ((void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              (void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              (void)hipFuncSetAttribute((const void*)aae::conv_first_multi_kernel<KS, C, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once;
    if (u8 && vec4) AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, true, true, false>), grid, dim3(256), smem, stream, m);
    else if (u8) AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, true, false, false>), grid, dim3(256), smem, stream, m);
    else AAE_LAUNCH((aae::conv_first_multi_kernel<KS, C, false, false, false>), grid, dim3(256), smem, stream, m);
}

template <int MT, int NT, bool SPREAD, int DEPTH = 2>
static void launch_wavek_multi_t(const aae::ConvWaveKMultiArgs& m, int tag, int nblk, hipStream_t stream) {
    constexpr int smem = aae::conv_wavek_smem<MT, NT, 4>();
    // TAG only makes the symbol unique per encoder layer (separate rows in rocprofv3 --stats)
    static const bool once = ((void)hipFuncSetAttribute((const void*)aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 0, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                              (void)hipFuncSetAttribute((const void*)aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 1, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                              (void)hipFuncSetAttribute((const void*)aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 2, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                              (void)hipFuncSetAttribute((const void*)aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 3, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem), true);
    (void)once;
    if (tag == 1) AAE_LAUNCH((aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 1, SPREAD>), dim3(nblk), dim3(256), smem, stream, m);
    else if (tag == 2) AAE_LAUNCH((aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 2, SPREAD>), dim3(nblk), dim3(256), smem, stream, m);
    else if (tag == 3) AAE_LAUNCH((aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 3, SPREAD>), dim3(nblk), dim3(256), smem, stream, m);
    else AAE_LAUNCH((aae::conv_wavek_multi_kernel<MT, NT, 4, DEPTH, 0, SPREAD>), dim3(nblk), dim3(256), smem, stream, m);
}

template <int RH>
static void launch_scan_resident_multi_t(const aae::ScanResidentMultiArgs& m, int nblk, hipStream_t stream) {
    constexpr int smem = aae::scan_resident_smem<false, RH>();
    static const bool once = ((void)hipFuncSetAttribute((const void*)aae::scan_resident_multi_kernel<RH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem), true);
    (void)once;
    AAE_LAUNCH((aae::scan_resident_multi_kernel<RH>), dim3(nblk), dim3(aae::kScanResidentThreads), smem, stream, m);
}

// The codebook scans of a mid-batch group: objects whose query takes the query-resident arg-max form on an fp32 codebook (the block normalises its own queries) share ONE
// launch per row-part count (plan_scan: res_rh by the query count) and ONE arg-max reduce launch -- every block the object's own launch's block, partials and answers in
// the object's own workspace slice: bit-identical.  Returns the members it did not cover in `rest`.
static int launch_mid_scans(const aae_multi_item* items, const MultiPlan& mp, const std::vector<int>& members, const float* z, int J, int64_t* idx_out, float* score_out,
                            unsigned char* base, hipStream_t stream, std::vector<int>& rest) {
    struct Member { int i; const aae_codebook* cb; ScanPlan s; int idx_scale; };
    std::vector<Member> ok;
    rest.clear();
    for (int i : members) {
        const aae_multi_item& it = items[i];
        const MultiItemPlan& p = mp.items[(size_t)i];
        const aae_codebook* cb = it.cb;
        int idx_scale = 1, col_stride = it.col_stride;
        if (col_stride > 1 && cb->upright && cb->upright_stride == col_stride) { idx_scale = col_stride; cb = cb->upright; col_stride = 1; }
        const ScanPlan s = plan_scan(cb, p.n, 1, col_stride > 1);
        const bool fits = it.enc->multi_mid_scan && col_stride == 1 && cb->dtype == AAE_DTYPE_F32 && s.resident_ok && !s.stream && cb->scan_fused_norm && !cb->scan_resident_fin &&
                          cb->scan_mode == AAE_SCAN_AUTO && (((uintptr_t)(z + (size_t)p.row0 * J)) & 15) == 0 && s.total <= p.cb_bytes;
        if (fits) ok.push_back({i, cb, s, idx_scale});
        else rest.push_back(i);
    }
    if (ok.size() < 2) {
        for (const Member& mbr : ok) rest.push_back(mbr.i);
        return AAE_OK;
    }
    for (int rh : {1, 2, 4}) {
        aae::ScanResidentMultiArgs m;
        memset(&m, 0, sizeof(m));
        int at = 0, nk = 0;
        for (const Member& mbr : ok) {
            if (mbr.s.res_rh != rh) continue;
            const MultiItemPlan& p = mp.items[(size_t)mbr.i];
            dim3 grid;
            m.item[nk] = scan_resident_args(mbr.cb, nullptr, p.n, mbr.s, base + p.cb_off, 1, z + (size_t)p.row0 * J, nullptr, &grid);
            m.row_blocks[nk] = (int)grid.x;
            m.range.first[nk] = at;
            at += (int)(grid.x * grid.y);
            ++nk;
        }
        if (!nk) continue;
        m.range.n = nk;
        m.range.first[nk] = at;
        if (rh == 1) launch_scan_resident_multi_t<1>(m, at, stream);
        else if (rh == 2) launch_scan_resident_multi_t<2>(m, at, stream);
        else launch_scan_resident_multi_t<4>(m, at, stream);
        AAE_HIP_TRY(hipGetLastError());
        ++t_multi_launches;
    }
    aae::ArgmaxReduceMultiArgs r;
    memset(&r, 0, sizeof(r));
    int at = 0, nk = 0;
    for (const Member& mbr : ok) {
        const MultiItemPlan& p = mp.items[(size_t)mbr.i];
        aae::ArgmaxReduceArgs& a = r.item[nk];
        a.pval = reinterpret_cast<float*>(base + p.cb_off + mbr.s.pval_off);
        a.pidx = reinterpret_cast<int*>(base + p.cb_off + mbr.s.pidx_off);
        a.idx_out = reinterpret_cast<long long*>(idx_out + p.row0);
        a.score_out = score_out + p.row0;
        a.nblk = mbr.s.res_blocks; a.B = p.n; a.Bstride = mbr.s.Bstride; a.idx_scale = mbr.idx_scale;
        r.range.first[nk] = at;
        at += p.n;
        ++nk;
    }
    r.range.n = nk;
    r.range.first[nk] = at;
    AAE_LAUNCH((aae::argmax_reduce_multi_kernel), dim3(at), dim3(256), 64, stream, r);
    AAE_HIP_TRY(hipGetLastError());
    ++t_multi_launches;
    return AAE_OK;
}

// the scan of up to kMultiMax grouped items in one launch (z: the items' raw latent codes, rows in item order)
static int launch_scan_multi(const MultiPlan& mp, const std::vector<int>& members, const float* z, int J, int64_t* idx_out, float* score_out,
                             unsigned char* base, unsigned nonce, hipStream_t stream) {
    aae::ScanMultiArgs m;
    memset(&m, 0, sizeof(m));
    int n = 1;                                  // NQ of the launch: the largest detection count among its objects (an object with fewer leaves rows idle;
    for (int i : members) n = std::max(n, mp.items[(size_t)i].n);      //  per-row scores and the arg-max do not depend on it)
    int at = 0;
    m.range.n = (int)members.size();
    for (size_t k = 0; k < members.size(); ++k) {
        const MultiItemPlan& p = mp.items[(size_t)members[k]];
        const aae_codebook* eff = p.eff;
        unsigned char* cbase = base + p.cb_off;
        aae::ScanArgs& a = m.item[k];
        a.E = eff->E; a.e_bytes = (unsigned)((size_t)eff->N * eff->J * sizeof(float));
        a.q = nullptr; a.qp = nullptr; a.cs = nullptr;
        a.z = z + (size_t)p.row0 * J;
        a.pval = reinterpret_cast<float*>(cbase + p.sp.pval_off);
        a.pidx = reinterpret_cast<int*>(cbase + p.sp.pidx_off);
        a.N = eff->N; a.J = eff->J; a.Jpad = p.sp.Jpad; a.B = p.n; a.Bpad = p.sp.Bpad; a.Bstride = p.sp.Bstride; a.col_stride = 1;
        a.tickets = reinterpret_cast<unsigned long long*>(cbase + p.sp.ticket_off); a.nonce = nonce;
        a.idx_out = reinterpret_cast<long long*>(idx_out + p.row0); a.score_out = score_out + p.row0; a.idx_scale = p.idx_scale;
        m.range.first[k] = at;
        at += p.sp.nblk;
    }
    m.range.first[members.size()] = at;
    const int smem = n * 128 * (int)sizeof(float) + aae::kScanTicketSmem;
    if (n == 1) AAE_LAUNCH((aae::scan_stream_multi_kernel<1>), dim3(at), dim3(256), smem, stream, m);
    else if (n == 2) AAE_LAUNCH((aae::scan_stream_multi_kernel<2>), dim3(at), dim3(256), smem, stream, m);
    else if (n == 3) AAE_LAUNCH((aae::scan_stream_multi_kernel<3>), dim3(at), dim3(256), smem, stream, m);
    else AAE_LAUNCH((aae::scan_stream_multi_kernel<4>), dim3(at), dim3(256), smem, stream, m);
    AAE_HIP_TRY(hipGetLastError());
    ++t_multi_launches;
    return AAE_OK;
}

// conv1 ... dense of up to kMultiMax grouped items (equal signatures): one launch per layer
static int launch_encoder_multi(const aae_multi_item* items, const MultiPlan& mp, const std::vector<int>& members, const std::vector<char>& wino_layers, const void* x, int x_dtype,
                                float* z_out, unsigned char* base, unsigned nonce, hipStream_t stream) {
    const aae_encoder* enc0 = items[members[0]].enc;
    const size_t nl = enc0->layers.size();
    const int J = enc0->desc.latent_size;
    int nmax = 1;                               // MQ of the GEMV launch: the largest detection count among the group's objects
    for (int i : members) nmax = std::max(nmax, mp.items[(size_t)i].n);
    const bool u8 = x_dtype == AAE_DTYPE_U8;
    const size_t crop_bytes = (size_t)enc0->desc.in_h * enc0->desc.in_w * enc0->desc.in_c * (u8 ? 1 : 4);
    auto enc_base = [&](size_t k) { return base + mp.items[(size_t)members[k]].enc_off; };
    auto tickets_of = [&](size_t k) { return reinterpret_cast<unsigned long long*>(enc_base(k) + mp.items[(size_t)members[k]].ws.ticket_off); };

    // ---- conv1 (+ one ticket-preparation block per object)
    {
        aae::ConvFirstMultiArgs m;
        memset(&m, 0, sizeof(m));
        m.range.n = (int)members.size();
        m.nonce = nonce;
        int at = 0, vec4 = -1;
        for (size_t k = 0; k < members.size(); ++k) {
            const aae_multi_item& it = items[members[k]];
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& L = it.enc->layers[0];
            const void* xk = static_cast<const unsigned char*>(x) + (size_t)p.row0 * crop_bytes;
            const int runs = first_core_args(it.enc, L, xk, u8, p.n, reinterpret_cast<float*>(enc_base(k) + p.ws.act_off[0]), false, m.item[k]);
            if (vec4 < 0) vec4 = m.item[k].vec4;
            if (m.item[k].vec4 != vec4) return fail(AAE_ERR_RUNTIME, "multi-object query: objects disagree on the dword staging of conv1 (option first_vec4)");
            m.range.first[k] = at;
            at += runs;
            aae::MultiTicketPrep& tp = m.prep[k];
            tp.n = 0;
            if (it.enc->ticket_prep) {
                auto add = [&](unsigned long long* words, int count) {
                    if (tp.n < aae::kMultiPrepRanges) { tp.words[tp.n] = words; tp.count[tp.n] = count; ++tp.n; }
                };
                for (size_t li = 1; li < nl; ++li)
                    if ((p.plans[li].gsplits > 1 || p.plans[li].tail_tiles > 0) && !(li < wino_layers.size() && wino_layers[li]))
                        add(tickets_of(k) + li * kLayerTicketWords, p.plans[li].tail_tiles > 0 ? p.plans[li].tail_tiles : p.plans[li].num_mt * p.plans[li].num_nt);
                add(tickets_of(k) + kConvTicketBytes / 8, (it.enc->dense.CoutPad / 128) * aae::kTicketSlotWords);
                add(reinterpret_cast<unsigned long long*>(base + p.cb_off + p.sp.ticket_off), aae::kTicketSlotWords);
            }
        }
        m.range.first[members.size()] = at;
        const Layer& L0 = enc0->layers[0];
        const dim3 grid(at + (unsigned)members.size(), ceil_div(L0.Cout, 128), 4);
        if (L0.Cin == 3) launch_first_multi_t<5, 3>(m, u8, vec4 != 0, grid, L0.first_smem, stream);
        else launch_first_multi_t<5, 1>(m, u8, vec4 != 0, grid, L0.first_smem, stream);
        AAE_HIP_TRY(hipGetLastError());
        ++t_multi_launches;
    }
    // ---- conv2 ...: the wave-split-K kernel, every object with its own plan -- or, where the group's blocks fill the chip, the Winograd layer kernel across the objects
    for (size_t li = 1; li < nl; ++li) {
        if (li < wino_layers.size() && wino_layers[li]) {
            const Layer& L0 = enc0->layers[li];
            aae::ConvWinoMultiArgs wm;
            memset(&wm, 0, sizeof(wm));
            aae::ConvWinoArgs& c = wm.c;
            c.H = L0.H; c.W = L0.W; c.Cin = L0.Cin; c.Cout = L0.Cout; c.Ho = L0.Ho; c.Wo = L0.Wo; c.relu = L0.relu;
            c.blocks_x = L0.wino_geom == 0 ? L0.Wo / 16 : 1;
            c.blocks_y = L0.wino_geom == 0 ? L0.Ho / 16 : 1;
            int wat = 0;
            wm.range.n = (int)members.size();
            for (size_t k = 0; k < members.size(); ++k) {
                const MultiItemPlan& p = mp.items[(size_t)members[k]];
                const Layer& L = items[members[k]].enc->layers[li];
                aae::ConvWinoObject& ob = wm.obj[k];
                ob.x = reinterpret_cast<const float*>(enc_base(k) + p.ws.act_off[li - 1]);
                ob.out = reinterpret_cast<float*>(enc_base(k) + p.ws.act_off[li]);
                for (int q = 0; q < 4; ++q) ob.U4[q] = L.wino[q];
                ob.bias = L.bias; ob.bn_scale = L.bn_scale; ob.bn_shift = L.bn_shift; ob.B = p.n;
                wm.range.first[k] = wat;
                wat += wino_regions(L, p.n);
            }
            wm.range.first[members.size()] = wat;
            c.regions = wat;
            c.xcd_cols = wino_xcd_cols(enc0, L0);
            wino_layer_multi_launch(L0.wino_geom, aae::wino_grid_blocks(wat, L0.Cout / 64, c.xcd_cols), stream, wm);
            AAE_HIP_TRY(hipGetLastError());
            ++t_multi_launches;
            continue;
        }
        aae::ConvWaveKMultiArgs m;
        memset(&m, 0, sizeof(m));
        m.range.n = (int)members.size();
        int at = 0;
        for (size_t k = 0; k < members.size(); ++k) {
            const aae_multi_item& it = items[members[k]];
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& L = it.enc->layers[li];
            const WaveKPlan& w = p.plans[li];
            m.item[k] = wavek_args(it.enc, L, w, reinterpret_cast<const float*>(enc_base(k) + p.ws.act_off[li - 1]), p.n * L.Ho * L.Wo,
                                   reinterpret_cast<float*>(enc_base(k) + p.ws.act_off[li]), reinterpret_cast<float*>(enc_base(k) + p.ws.partial_off),
                                   tickets_of(k) + li * kLayerTicketWords, nonce, (int)li);
            m.item[k].timeline = nullptr;
            if (wavek_shape_key(w) != wavek_shape_key(mp.items[(size_t)members[0]].plans[li]))       // (one kernel instantiation serves the launch)
                return fail(AAE_ERR_RUNTIME, "multi-object query: the members of a group disagree on the wave tile of conv%zu", li + 1);
            m.nblk[k] = w.blocks();
            m.range.first[k] = at;
            at += (w.blocks() + 7) / 8 * 8;              // (every object's first block on XCD 0: xcd_remap counts from it)
        }
        m.range.first[members.size()] = at;
        m.xcd_affine = 0;
        if (enc0->multi_xcd_affine && members.size() % 8 == 0) {          // equal-sized objects, a multiple of 8 of them: one XCD per object
            const int nb = m.range.first[1] - m.range.first[0];
            bool equal = true;
            for (size_t k = 0; k < members.size(); ++k) equal = equal && m.range.first[k + 1] - m.range.first[k] == nb;
            if (equal) m.xcd_affine = nb;
        }
        const WaveKPlan& w0 = mp.items[(size_t)members[0]].plans[li];
        const int tag = li <= 3 ? (int)li : 0;
        switch (wavek_shape_key(w0)) {
            case 1142: launch_wavek_multi_t<1, 1, true>(m, tag, at, stream); break;
            case 142:
#ifdef AAE_EXPERIMENTS
                // (A/B: the spread load schedule for 64 x 32 tiles -- loads between the MFMAs of its two accumulators: 8 x 1 434 against 400 us,
                //  16 x 1 807 against 743: slower, as on single objects in round 4; profiles/r13_multi/depth_and_schedule_ab.jsonl)
                if (enc0->multi_force_depth & 0x10000) { launch_wavek_multi_t<2, 1, true>(m, tag, at, stream); break; }
#endif
                launch_wavek_multi_t<2, 1, false>(m, tag, at, stream); break;
#ifdef AAE_EXPERIMENTS
            // three slabs in flight for the 64 x 32 layers (option multi_force_depth): conv4 of a few-detection frame streams 26 MB of cold weights
            // per object against 5 us of MFMA work -- more bytes in flight changed NOTHING (8 x 1: 397 vs 397 us, profiles/r13_multi/depth_ab.jsonl)
            case 143: launch_wavek_multi_t<2, 1, false, 3>(m, tag, at, stream); break;
#endif
            case 242: launch_wavek_multi_t<2, 2, true>(m, tag, at, stream); break;
            default: return fail(AAE_ERR_RUNTIME, "multi-object query: no grouped wave-split-K instantiation for shape key %d", wavek_shape_key(w0));
        }
        AAE_HIP_TRY(hipGetLastError());
        ++t_multi_launches;
    }
    // ---- dense layer: the ticketed GEMV
    {
        aae::DenseGemvMultiArgs m;
        memset(&m, 0, sizeof(m));
        m.range.n = (int)members.size();
        int at = 0;
        for (size_t k = 0; k < members.size(); ++k) {
            const aae_multi_item& it = items[members[k]];
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& D = it.enc->dense;
            aae::DenseGemvArgs a = gemv_args(D, reinterpret_cast<const float*>(enc_base(k) + p.ws.act_off[nl - 1]), p.n, reinterpret_cast<float*>(enc_base(k) + p.ws.partial_off));
            a.bias = D.bias; a.bn_scale = D.bn_scale; a.bn_shift = D.bn_shift; a.out = z_out + (size_t)p.row0 * J;
            a.tickets = tickets_of(k) + kConvTicketBytes / 8; a.nonce = nonce; a.relu = D.relu;
            m.item[k] = a;
            m.range.first[k] = at;
            at += ceil_div(a.K, aae::kGemvChunk);
        }
        m.range.first[members.size()] = at;
        const Layer& D0 = enc0->dense;
        const dim3 grid(at, D0.CoutPad / 128);
        int smem = 2 * nmax * aae::kGemvChunk * (int)sizeof(float) + 16;
        if (smem < aae::kGemvTicketSmem) smem = aae::kGemvTicketSmem;
        if (nmax == 1) AAE_LAUNCH((aae::dense_gemv_multi_kernel<1>), grid, dim3(256), smem, stream, m);
        else if (nmax == 2) AAE_LAUNCH((aae::dense_gemv_multi_kernel<2>), grid, dim3(256), smem, stream, m);
        else if (nmax == 3) AAE_LAUNCH((aae::dense_gemv_multi_kernel<3>), grid, dim3(256), smem, stream, m);
        else AAE_LAUNCH((aae::dense_gemv_multi_kernel<4>), grid, dim3(256), smem, stream, m);
        AAE_HIP_TRY(hipGetLastError());
        ++t_multi_launches;
    }
    return AAE_OK;
}

// A mid-batch group: conv1, every Winograd conv layer and the dense layer as ONE launch each across the objects, the scans in shared launches (launch_mid_scans).
static int launch_mid_group(const aae_multi_item* items, const MultiPlan& mp, const std::vector<int>& members, const std::vector<char>& wino_layers, const std::vector<char>& rem_layers, const void* x, int x_dtype,
                            float* z_out, int64_t* idx_out, float* score_out, unsigned char* base, void* stream_v) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    aae_encoder* enc0 = items[members[0]].enc;
    const size_t nl = enc0->layers.size();
    const int J = enc0->desc.latent_size;
    const size_t crop_bytes = (size_t)enc0->desc.in_h * enc0->desc.in_w * enc0->desc.in_c * (x_dtype == AAE_DTYPE_U8 ? 1 : 4);
    Timer tm;
    // ---- conv1: one launch across the objects where every member's first layer is the MFMA form with the same staging (whole 128-pixel tiles per block: every block
    //      exactly the object's own launch's block); otherwise per object
    bool conv1_grouped = true;
    {
        const bool u8 = x_dtype == AAE_DTYPE_U8;
        aae::ConvFirstMultiArgs m;
        memset(&m, 0, sizeof(m));
        m.range.n = (int)members.size();
        int at = 0, vec4 = -1;
        for (size_t k = 0; k < members.size() && conv1_grouped; ++k) {
            const aae_multi_item& it = items[members[k]];
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& L = it.enc->layers[0];
            if (L.kind != KIND_FIRST_MFMA || !(L.Cin == 3 || L.Cin == 1) || L.KS != 5 || L.first_smem != enc0->layers[0].first_smem || L.Cout != enc0->layers[0].Cout) { conv1_grouped = false; break; }
            const void* xk = static_cast<const unsigned char*>(x) + (size_t)p.row0 * crop_bytes;
            const int runs = first_core_args(it.enc, L, xk, u8, p.n, reinterpret_cast<float*>(base + p.enc_off + p.ws.act_off[0]), false, m.item[k]);
            if (m.item[k].total_tiles <= it.enc->first_group_split_max_tiles) conv1_grouped = false;       // (its own launch would take the group-split form: keep the bits)
            if (vec4 < 0) vec4 = m.item[k].vec4;
            if (m.item[k].vec4 != vec4) conv1_grouped = false;
            m.range.first[k] = at;
            at += runs;
        }
        if (conv1_grouped) {
            m.range.first[members.size()] = at;
            const Layer& L0 = enc0->layers[0];
            const dim3 grid((unsigned)at, ceil_div(L0.Cout, 128), 1);
            if (L0.Cin == 3) launch_first_multi_tiles_t<5, 3>(m, u8, vec4 != 0, grid, L0.first_smem, stream);
            else launch_first_multi_tiles_t<5, 1>(m, u8, vec4 != 0, grid, L0.first_smem, stream);
            AAE_HIP_TRY(hipGetLastError());
            ++t_multi_launches;
        }
    }
    if (!conv1_grouped)
        for (int i : members) {
            const MultiItemPlan& p = mp.items[(size_t)i];
            if (int rc = forward_impl(items[i].enc, static_cast<const unsigned char*>(x) + (size_t)p.row0 * crop_bytes, x_dtype, p.n, z_out + (size_t)p.row0 * J,
                                      base + p.enc_off, p.enc_bytes, stream_v, tm, nullptr, nullptr, nullptr, 0, 1)) return rc;
        }
    for (size_t li = 1; li < nl; ++li) {
        const Layer& L0 = enc0->layers[li];
        if (!(li < wino_layers.size() && wino_layers[li])) {
            // the group does not fill the chip on this layer: every object runs it alone, on the kernel its own forward takes at this batch
            for (int i : members) {
                const MultiItemPlan& p = mp.items[(size_t)i];
                if (int rc = forward_impl(items[i].enc, base + p.enc_off + p.ws.act_off[li - 1], AAE_DTYPE_F32, p.n, z_out + (size_t)p.row0 * J, base + p.enc_off, p.enc_bytes,
                                          stream_v, tm, nullptr, nullptr, nullptr, (int)li, (int)li + 1)) return rc;
            }
            continue;
        }
        aae::ConvWinoMultiArgs m;
        memset(&m, 0, sizeof(m));
        aae::ConvWinoArgs& c = m.c;
        c.H = L0.H; c.W = L0.W; c.Cin = L0.Cin; c.Cout = L0.Cout; c.Ho = L0.Ho; c.Wo = L0.Wo; c.relu = L0.relu;
        c.blocks_x = L0.wino_geom == 0 ? L0.Wo / 16 : 1;
        c.blocks_y = L0.wino_geom == 0 ? L0.Ho / 16 : 1;
        int at = 0;
        m.range.n = (int)members.size();
        const bool split = li < rem_layers.size() && rem_layers[li];
        for (size_t k = 0; k < members.size(); ++k) {
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& L = items[members[k]].enc->layers[li];
            aae::ConvWinoObject& ob = m.obj[k];
            ob.x = reinterpret_cast<const float*>(base + p.enc_off + p.ws.act_off[li - 1]);
            ob.out = reinterpret_cast<float*>(base + p.enc_off + p.ws.act_off[li]);
            for (int q = 0; q < 4; ++q) ob.U4[q] = L.wino[q];
            ob.bias = L.bias; ob.bn_scale = L.bn_scale; ob.bn_shift = L.bn_shift;
            ob.B = split ? p.n / 4 * 4 : p.n;                                  // (split: the complete four-image blocks only)
            m.range.first[k] = at;
            at += wino_regions(L, ob.B);
        }
        m.range.first[members.size()] = at;
        c.regions = at;
        c.xcd_cols = wino_xcd_cols(enc0, L0);
        wino_layer_multi_launch(L0.wino_geom, aae::wino_grid_blocks(at, L0.Cout / 64, c.xcd_cols), stream, m);
        AAE_HIP_TRY(hipGetLastError());
        ++t_multi_launches;
        if (split) {
            // the objects' last n mod 4 images: one grouped wave-split-K launch (plan_mid_ragged), every object its own tiles, tickets and partial sums
            aae::ConvWaveKMultiArgs r;
            memset(&r, 0, sizeof(r));
            const unsigned nonce = next_nonce();
            int rat = 0, nk = 0;
            const WaveKPlan* w0 = nullptr;
            for (size_t k = 0; k < members.size(); ++k) {
                const aae_multi_item& it = items[members[k]];
                const MultiItemPlan& p = mp.items[(size_t)members[k]];
                const Layer& L = it.enc->layers[li];
                const WaveKPlan& w = p.rem_plans[li];
                const int full = p.n / 4 * 4, rem = p.n - full;
                if (!rem || !w.use) continue;
                if (!w0) w0 = &w;
                unsigned long long* tickets = reinterpret_cast<unsigned long long*>(base + p.enc_off + p.ws.ticket_off) + li * kLayerTicketWords;
                r.item[nk] = wavek_args(it.enc, L, w, reinterpret_cast<const float*>(base + p.enc_off + p.ws.act_off[li - 1]) + (size_t)full * L.H * L.W * L.Cin, rem * L.Ho * L.Wo,
                                        reinterpret_cast<float*>(base + p.enc_off + p.ws.act_off[li]) + (size_t)full * L.Ho * L.Wo * L.Cout,
                                        reinterpret_cast<float*>(base + p.enc_off + p.rem_partial_off), tickets, nonce, (int)li);
                r.item[nk].timeline = nullptr;
                r.nblk[nk] = w.blocks();
                r.range.first[nk] = rat;
                rat += (w.blocks() + 7) / 8 * 8;
                ++nk;
            }
            r.range.n = nk;
            r.range.first[nk] = rat;
            r.xcd_affine = 0;
            const int tag = li <= 3 ? (int)li : 0;
            switch (w0 ? wavek_shape_key(*w0) : 0) {
                case 1142: launch_wavek_multi_t<1, 1, true>(r, tag, rat, stream); break;
                case 142: launch_wavek_multi_t<2, 1, false>(r, tag, rat, stream); break;
                case 242: launch_wavek_multi_t<2, 2, true>(r, tag, rat, stream); break;
                default: return fail(AAE_ERR_RUNTIME, "multi-object query: no grouped wave-split-K instantiation for the incomplete blocks of conv%zu", li + 1);
            }
            AAE_HIP_TRY(hipGetLastError());
            ++t_multi_launches;
        }
    }
    // ---- dense layer: one wave-split-K launch across the objects where every member's own plan is the same instantiated wave tile (each object its own plan, tickets and
    //      partial buffer: bit-identical to its own launch); otherwise per object
    bool dense_grouped = true;
    {
        aae::ConvWaveKMultiArgs m;
        memset(&m, 0, sizeof(m));
        m.range.n = (int)members.size();
        const unsigned nonce = next_nonce();
        int at = 0, key0 = -1;
        for (size_t k = 0; k < members.size() && dense_grouped; ++k) {
            const aae_multi_item& it = items[members[k]];
            const MultiItemPlan& p = mp.items[(size_t)members[k]];
            const Layer& D = it.enc->dense;
            const bool gemv = D.kind == KIND_IGEMM && p.n <= gemv_max_batch(it.enc) && it.enc->dense_gemv && D.K() % aae::kGemvChunk == 0;
            if (D.kind != KIND_IGEMM || gemv || !it.enc->wavek_dense) { dense_grouped = false; break; }
            const WaveKPlan w = plan_wavek(it.enc, D, p.n, false);
            const int key = wavek_shape_key(w);
            if (!w.use || w.waves != 4 || w.depth != 2 || w.tail_tiles > 0 || w.partial_bytes > p.ws.partial_bytes ||
                !((key == 1142 && (it.enc->wavek_spread & 2)) || key == 142 || (key == 242 && (it.enc->wavek_spread & 1)))) { dense_grouped = false; break; }
            if (key0 < 0) key0 = key;
            if (key != key0) { dense_grouped = false; break; }
            unsigned long long* tickets = reinterpret_cast<unsigned long long*>(base + p.enc_off + p.ws.ticket_off) + nl * kLayerTicketWords;
            m.item[k] = wavek_args(it.enc, D, w, reinterpret_cast<const float*>(base + p.enc_off + p.ws.act_off[nl - 1]), p.n, z_out + (size_t)p.row0 * J,
                                   reinterpret_cast<float*>(base + p.enc_off + p.ws.partial_off), tickets, nonce, 0);
            m.item[k].timeline = nullptr;
            m.nblk[k] = w.blocks();
            m.range.first[k] = at;
            at += (w.blocks() + 7) / 8 * 8;
        }
        if (dense_grouped) {
            m.range.first[members.size()] = at;
            m.xcd_affine = 0;
            switch (key0) {
                case 1142: launch_wavek_multi_t<1, 1, true>(m, 0, at, stream); break;
                case 142: launch_wavek_multi_t<2, 1, false>(m, 0, at, stream); break;
                case 242: launch_wavek_multi_t<2, 2, true>(m, 0, at, stream); break;
                default: dense_grouped = false;
            }
            if (dense_grouped) {
                AAE_HIP_TRY(hipGetLastError());
                ++t_multi_launches;
            }
        }
    }
    if (!dense_grouped)
        for (int i : members) {
            const MultiItemPlan& p = mp.items[(size_t)i];
            if (int rc = forward_impl(items[i].enc, base + p.enc_off + p.ws.act_off[nl - 1], AAE_DTYPE_F32, p.n, z_out + (size_t)p.row0 * J, base + p.enc_off, p.enc_bytes,
                                      stream_v, tm, nullptr, nullptr, nullptr, (int)nl, (int)nl + 1)) return rc;
        }
    std::vector<int> rest;
    if (int rc = launch_mid_scans(items, mp, members, z_out, J, idx_out, score_out, base, stream, rest)) return rc;
    for (int i : rest) {
        const MultiItemPlan& p = mp.items[(size_t)i];
        if (int rc = aae_codebook_nn(items[i].cb, z_out + (size_t)p.row0 * J, p.n, 1, items[i].col_stride, idx_out + p.row0, score_out + p.row0, base + p.cb_off,
                                     p.cb_bytes, stream_v)) return rc;
    }
    return AAE_OK;
}

// A class with a handful of boxes beyond four is cheapest INSIDE the frame's per-detection group: as consecutive items of at most four boxes that share the class's handles (rows stay in
// order).  Measured against the mid-batch group / the class's own call (tools/split_items_ab.py, profiles/r15/split_items_ab.jsonl): 4 x 5 boxes -12 %, 4 x 8 -13 %, 8 x 6 -4 %,
// 8 classes x {9,1,1,1,1,1,1,1} -23 %; from 10 boxes per class on the whole class wins (4 x 10 +9 %, 8 x 16 +20 % when split).  Rule: split when EVERY class of the frame beyond four boxes has
// 5 ... 8 (up to 12 when it is the only one); never past kMultiMax items in all, never without the group plan (multi_group_plan = 0 promises the per-object bits).
static bool expand_items(const aae_multi_item* items, int n_items, bool scan_only, std::vector<aae_multi_item>& out) {
    if (scan_only || !items || n_items < 1) return false;
    int beyond4 = 0, total = 0;
    for (int i = 0; i < n_items; ++i) {
        if (!items[i].enc || !items[i].cb || items[i].n < 1) return false;
        if (items[i].n > 4) ++beyond4;
    }
    if (!beyond4) return false;
    std::vector<int> parts((size_t)n_items, 1);
    bool any = false;
    for (int i = 0; i < n_items; ++i) {
        const aae_multi_item& it = items[i];
        const aae_encoder* e = it.enc;
        std::vector<WaveKPlan> plans;
        std::vector<int> sig;
        const int limit = beyond4 == 1 ? 12 : 8;
        const aae_codebook* eff = nullptr;
        int idx_scale = 1;
        if (it.n > 4 && it.n <= limit && e->multi_split_items && e->multi_group_plan && multi_encoder_groupable(e, 4, plans, sig) &&
            multi_scan_groupable(it.cb, 4, it.col_stride, &eff, &idx_scale)) {
            parts[(size_t)i] = ceil_div(it.n, 4);
            any = true;
        } else if (it.n > 4) {
            return false;          // (a larger class stays whole -- then the small ones are worth more as its partners in a mid-batch group: {5, 9, 14} split only in part +19 %)
        }
        total += parts[(size_t)i];
    }
    if (!any || total > aae::kMultiMax) return false;
    out.clear();
    for (int i = 0; i < n_items; ++i) {
        int left = items[i].n;
        for (int k = 0; k < parts[(size_t)i]; ++k) {
            aae_multi_item sub = items[i];
            sub.n = parts[(size_t)i] == 1 ? left : std::min(4, left);
            left -= sub.n;
            out.push_back(sub);
        }
    }
    return true;
}

static int multi_impl(const aae_multi_item* items_in, int n_items_in, const void* x, int x_dtype, const float* z_in, float* z_out, int64_t* idx_out,
                      float* score_out, void* workspace, size_t ws_bytes, void* stream_v) {
    const bool scan_only = z_in != nullptr;
    std::vector<aae_multi_item> expanded;
    const bool split = expand_items(items_in, n_items_in, scan_only, expanded);
    const aae_multi_item* items = split ? expanded.data() : items_in;
    const int n_items = split ? (int)expanded.size() : n_items_in;
    t_multi_launches = 0;
    MultiPlan mp;
    if (int rc = plan_multi(items, n_items, scan_only, mp)) return rc;
    if (!idx_out || !score_out || (!scan_only && (!x || !z_out))) return fail(AAE_ERR_INVALID, "multi-object query: null argument");
    if (!scan_only && x_dtype != AAE_DTYPE_U8 && x_dtype != AAE_DTYPE_F32)
        return fail(AAE_ERR_INVALID, "multi-object query: x_dtype %d (want AAE_DTYPE_U8 or AAE_DTYPE_F32)", x_dtype);
    if (ws_bytes < mp.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B (aae_multi_workspace_bytes)", ws_bytes, mp.total);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    const int J = items[0].cb->J;
    const float* z = scan_only ? z_in : z_out;
    // ---- the items the grouped kernels do not cover: the per-object path, one after the other in the shared slice
    for (int i = 0; i < n_items; ++i) {
        const MultiItemPlan& p = mp.items[(size_t)i];
        if (p.grouped || p.mid) continue;
        const aae_multi_item& it = items[i];
        int rc;
        if (scan_only)
            rc = aae_codebook_nn(it.cb, z + (size_t)p.row0 * J, it.n, 1, it.col_stride, idx_out + p.row0, score_out + p.row0, base + mp.seq_cb_off, mp.seq_cb_bytes, stream_v);
        else {
            const size_t crop_bytes = (size_t)it.enc->desc.in_h * it.enc->desc.in_w * it.enc->desc.in_c * (x_dtype == AAE_DTYPE_U8 ? 1 : 4);
            if (crop_bytes != (size_t)items[0].enc->desc.in_h * items[0].enc->desc.in_w * items[0].enc->desc.in_c * (x_dtype == AAE_DTYPE_U8 ? 1 : 4))
                return fail(AAE_ERR_INVALID, "multi-object query: item %d takes crops of another shape than item 0 (one [rows, H, W, C] crop array serves all)", i);
            rc = aae_encode_nn(it.enc, it.cb, static_cast<const unsigned char*>(x) + (size_t)p.row0 * crop_bytes, x_dtype, it.n, it.col_stride,
                               z_out + (size_t)p.row0 * J, idx_out + p.row0, score_out + p.row0, base + mp.seq_enc_off, mp.seq_enc_bytes,
                               base + mp.seq_cb_off, mp.seq_cb_bytes, stream_v);
        }
        if (rc) return rc;
    }
    // ---- mid-batch groups: one Winograd launch per conv layer and group
    for (size_t gi = 0; gi < mp.mid_groups.size(); ++gi) {
        const std::vector<int>& g = mp.mid_groups[gi];
        for (int i : g) {
            const aae_encoder_desc &a = items[i].enc->desc, &b = items[g[0]].enc->desc;
            if (a.in_h != b.in_h || a.in_w != b.in_w || a.in_c != b.in_c) return fail(AAE_ERR_RUNTIME, "multi-object query: group members differ in crop shape");
        }
        static const std::vector<char> none;
        if (int rc = launch_mid_group(items, mp, g, gi < mp.mid_wino.size() ? mp.mid_wino[gi] : none, gi < mp.mid_rem.size() ? mp.mid_rem[gi] : none, x, x_dtype, z_out, idx_out,
                                      score_out, base, stream_v)) return rc;
    }
    // ---- grouped items: one launch per layer and group
    if (!mp.groups.empty()) t_x3h_last_slot = -1;
    for (size_t gidx = 0; gidx < mp.groups.size(); ++gidx) {
        const std::vector<int>& g = mp.groups[gidx];
        const unsigned nonce = next_nonce();           // one per group and call: every ticketed launch has its own words
        if (!scan_only) {
            for (int i : g) {
                const aae_encoder_desc &a = items[i].enc->desc, &b = items[g[0]].enc->desc;
                if (a.in_h != b.in_h || a.in_w != b.in_w || a.in_c != b.in_c) return fail(AAE_ERR_RUNTIME, "multi-object query: group members differ in crop shape");
            }
            static const std::vector<char> no_wino;
            if (int rc = launch_encoder_multi(items, mp, g, gidx < mp.group_wino.size() ? mp.group_wino[gidx] : no_wino, x, x_dtype, z_out, base, nonce, stream)) return rc;
        }
        if (int rc = launch_scan_multi(mp, g, z, J, idx_out, score_out, base, nonce, stream)) return rc;
    }
    return AAE_OK;
}

}  // namespace aae_host

extern "C" {

size_t aae_multi_workspace_bytes(const aae_multi_item* items_in, int n_items_in, int scan_only) {
    std::vector<aae_multi_item> expanded;
    const bool split = aae_host::expand_items(items_in, n_items_in, scan_only != 0, expanded);
    const aae_multi_item* items = split ? expanded.data() : items_in;
    const int n_items = split ? (int)expanded.size() : n_items_in;
    // (the one place outside the hot calls that sees a frame's layout: objects that may join a mid-batch group get their Winograd weights here)
    // Winograd-domain weights (+83.5 MB for the reference network) only for the objects of a group that WOULD form (an estimator with thirty classes of a few boxes each never builds them).
    if (items && !scan_only) {
        for (int i = 0; i < n_items; ++i)          // (an object whose OWN forward at this count takes the Winograd form: as aae_encoder_workspace_bytes does)
            if (items[i].enc && items[i].n >= 1 && aae_host::wants_winograd_weights(items[i].enc, items[i].n) && aae_host::ensure_winograd_weights(items[i].enc) != AAE_OK) return 0;
        {   // per-detection groups with a layer in the Winograd form (multi_group_winograd): a dry plan tells which
            aae_host::MultiPlan dry;
            if (aae_host::plan_multi(items, n_items, false, dry, true) != AAE_OK) return 0;
            for (size_t gi = 0; gi < dry.groups.size(); ++gi) {
                bool any = false;
                for (char c : dry.group_wino[gi]) any = any || c;
                if (any)
                    for (int k : dry.groups[gi])
                        if (aae_host::ensure_winograd_weights(items[k].enc) != AAE_OK) return 0;
            }
        }
        std::vector<std::vector<int>> sigs((size_t)n_items);
        std::vector<char> cand((size_t)n_items, 0), seen((size_t)n_items, 0);
        for (int i = 0; i < n_items; ++i)
            cand[(size_t)i] = items[i].enc && items[i].cb && items[i].n >= 5 && aae_host::multi_encoder_mid_groupable(items[i].enc, items[i].n, sigs[(size_t)i], true);
        for (int i = 0; i < n_items; ++i) {
            if (!cand[(size_t)i] || seen[(size_t)i]) continue;
            std::vector<int> members, counts;
            for (int k = i; k < n_items && (int)members.size() < aae::kMultiMax; ++k)
                if (cand[(size_t)k] && !seen[(size_t)k] && sigs[(size_t)k] == sigs[(size_t)i]) { members.push_back(k); counts.push_back(items[k].n); seen[(size_t)k] = 1; }
            if (members.size() >= 2 && aae_host::mid_group_fills(items[members[0]].enc, counts))
                for (int k : members)
                    if (aae_host::ensure_winograd_weights(items[k].enc) != AAE_OK) return 0;
        }
    }
    aae_host::MultiPlan mp;
    if (aae_host::plan_multi(items, n_items, scan_only != 0, mp) != AAE_OK) return 0;
    return mp.total;
}

int aae_multi_rows(const aae_multi_item* items, int n_items) {
    if (!items || n_items < 1) return 0;
    long long rows = 0;
    for (int i = 0; i < n_items; ++i) rows += items[i].n > 0 ? items[i].n : 0;
    return rows > 0x7fffffff ? 0 : (int)rows;
}

int aae_encode_nn_multi(const aae_multi_item* items, int n_items, const void* x, int x_dtype, float* z_out, int64_t* idx_out, float* score_out,
                        void* workspace, size_t ws_bytes, void* stream) {
    return aae_host::multi_impl(items, n_items, x, x_dtype, nullptr, z_out, idx_out, score_out, workspace, ws_bytes, stream);
}

int aae_codebook_nn_multi(const aae_multi_item* items, int n_items, const float* z, int64_t* idx_out, float* score_out, void* workspace,
                          size_t ws_bytes, void* stream) {
    if (!z) return aae_host::fail(AAE_ERR_INVALID, "aae_codebook_nn_multi: null latent array");
    return aae_host::multi_impl(items, n_items, nullptr, AAE_DTYPE_F32, z, nullptr, idx_out, score_out, workspace, ws_bytes, stream);
}

int aae_detect_nn_multi(const aae_multi_item* items, int n_items, const void* img, int H, int W, int C, const int32_t* boxes, void* crops,
                        float* z_out, int64_t* idx_out, float* score_out, void* workspace, size_t ws_bytes, void* stream) {
    using namespace aae_host;
    if (!items || n_items < 1 || !items[0].enc || !crops) return fail(AAE_ERR_INVALID, "aae_detect_nn_multi: null argument");
    const aae_encoder_desc& d = items[0].enc->desc;
    if (C != d.in_c) return fail(AAE_ERR_INVALID, "aae_detect_nn_multi: image has %d channels, the encoders take %d", C, d.in_c);
    for (int i = 1; i < n_items; ++i)
        if (!items[i].enc || items[i].enc->desc.in_h != d.in_h || items[i].enc->desc.in_w != d.in_w || items[i].enc->desc.in_c != d.in_c)
            return fail(AAE_ERR_INVALID, "aae_detect_nn_multi: item %d takes crops of another shape than item 0 (one crop array serves all)", i);
    const int rows = aae_multi_rows(items, n_items);
    if (rows < 1) return fail(AAE_ERR_INVALID, "aae_detect_nn_multi: no detections");
    for (int a = 0; a < rows; a += 65535) {              // (the crop kernel's grid.y)
        const int m = std::min(65535, rows - a);
        if (int rc = aae_crop_resize_u8(img, H, W, C, boxes + (size_t)a * 5, m, d.in_h, d.in_w,
                                        static_cast<unsigned char*>(crops) + (size_t)a * d.in_h * d.in_w * d.in_c, stream)) return rc;
    }
    return aae_encode_nn_multi(items, n_items, crops, AAE_DTYPE_U8, z_out, idx_out, score_out, workspace, ws_bytes, stream);
}

int aae_multi_last_launches(void) { return aae_host::t_multi_launches; }

}  // extern "C"
