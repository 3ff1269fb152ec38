/* aae_hip_tuning.h -- the launch-planning knobs and A/B switches of libaae_hip.so.
 *
 * Nothing here is needed to USE the library: include/aae_hip.h documents the options a caller may want ("winograd", "precision",
 * "x3h_act_shift", "compact_workspace", "dense_gemv", "multi_group_plan") and the scan modes AAE_SCAN_AUTO /
 * AAE_SCAN_AUTO_PACKED.  This header lists everything else aae_encoder_set_option() / aae_codebook_set_scan_mode() accept --
 * the constants the planner was tuned with and the switches the A/B measurements of CHANGELOG.md were taken with -- and
 * which of them exist in which build:
 *
 *   product build      libaae_hip.so              the kernel forms the planner uses.  Every accepted option value gives
 *                                                 results that are bit-identical to the defaults or differ from them by fp32
 *                                                 summation order only; NO option makes results wrong.
 *   experiments build  libaae_hip_experiments.so  the same sources with -DAAE_EXPERIMENTS (`python __graft_entry__.py
 *                                                 experiments`; AAE_EXPERIMENTS=1 in the environment makes the Python mirror
 *                                                 load it): additionally every kernel variant that measured SLOWER than the
 *                                                 default it was built against, and the profiling / ablation aids.  Used by
 *                                                 tools/ and by the A/B tests; aae_has_experiments() returns 1.
 *
 * In the product build an experiments-only option accepts its default value and answers AAE_ERR_UNSUPPORTED to any other.
 */
#ifndef AAE_HIP_TUNING_H_
#define AAE_HIP_TUNING_H_

#include "aae_hip.h"

/* ---- scan modes beyond AAE_SCAN_AUTO / AAE_SCAN_AUTO_PACKED (aae_codebook_set_scan_mode); every mode returns the same answers ---- */
#define AAE_SCAN_GEMV 1           /* [experiments] round-1 vector-ALU scan with shuffle reductions, B <= 4                          */
#define AAE_SCAN_MFMA 2           /* tile-resident matrix-core kernels at any B (what similarity / masked upright queries use)       */
#define AAE_SCAN_STREAM 3         /* the B <= 4 stream kernel with its in-launch finish, forced (AUTO picks it)                      */
#define AAE_SCAN_STREAM_2L 4      /* ... followed by a separate arg-max reduce launch instead (A/B, race screen)                     */
#define AAE_SCAN_AUTO_NO_PRUNE 5  /* AUTO with un-pruned top-k lists inside the query-resident scan (A/B of the shared bound)        */
#define AAE_SCAN_STREAM_WALK 6    /* [experiments] one block per CU walking the codebook, two 32-row batches in flight per wave:
                                     level at B = 1, slower at B = 4 (CHANGELOG.md round 4)                                          */
#define AAE_SCAN_AUTO_RH2 8       /* AUTO, but <= 32 queries split a tile's rows over two waves per query group instead of four      */
#define AAE_SCAN_AUTO_FIN 9       /* AUTO, and the <= 32-query resident scan answers inside its launch (ticket finish): measured
                                     slower than the reduce launch it replaces (20.7 vs 16.1 us), opt-in                             */

/* ---- encoder options (aae_encoder_set_option(enc, name, value)); defaults in parentheses ---------------------------------------
 * large batches (128-row implicit GEMM, conv_igemm_f32.h / conv_igemm_x3h.h)
 *   "splitk_min_base_blocks" (384), "splitk_target_blocks" (512): split K only when the un-split grid is smaller / aim for this many blocks
 *   "igemm_breg_wide" (1), "igemm_breg_wide_min_blocks" (512), "igemm_breg_min_blocks" (768): 128 x 256 block tiles / LDS footprint choice
 *   "igemm_stagger" (0): start delay (kcycles) for every 2nd block generation -- measured neutral
 *   "reduce_small" (1): <= 8 splits over >= 16k outputs by the barrier-free float4 reduce
 *   "x3h_wide256" (1), "x3h_wide256_min_blocks" (256), "x3h_min_tiles" (256): f32x3h tile shape / where precision 2 switches over
 *   "first_target_blocks" (512), "first_max_tiles_per_block" (16), "first_vec4" (1), "first_group_split_max_tiles" (128): conv1 grid shaping
 *   [experiments] "igemm_dma" (1), "igemm_breg" (1), "x3h_dma" (1): 0 = the register-staged operand paths (bit-identical, slower)
 *   [experiments] "x3h_wide_min_blocks" (0): > 0 = 256 x 128 f32x3h tiles (measured neutral)
 * polyphase Winograd conv layers (conv_winograd_f32.h; "winograd" itself: include/aae_hip.h)
 *   "winograd_min_batch" (8), "winograd_min_fill_pct" (56): a layer takes the Winograd form when its 64-tile x 64-channel blocks fill at least
 *                        this share of the rounds of blocks (one per compute unit) they occupy; "winograd_min_blocks" (0): > 0 = a plain block count instead
 *   "winograd_xcd_cols" (-1): 64-column blocks of a window region that share an XCD (-1 = per layer: all of them up to four; 0 = plain block order)
 *   "multi_split_items" (1): aae_encode_nn_multi -- a class with 5 ... 8 boxes in the frame (up to 12 when no other class has more than 4) joins the per-detection group as items of
 *                        at most 4 boxes; 0 = such a class is a mid-batch candidate / its own call.  Needs "multi_group_plan" = 1
 *   "multi_group_winograd" (1): aae_encode_nn_multi -- in a group of objects with 1 ... 4 detections each a conv layer runs as one Winograd launch across the objects where the
 *                        group's blocks fill the chip (conv2 from ~9 detections per frame, conv3 from ~18); needs "multi_group_plan" = 1
 *   "multi_mid_scan" (1): ... and the codebook scans of the group's objects share their launches (bit-identical answers); 0 = one scan + one reduce launch per object
 *   "multi_mid_ragged" (1): ... and where the objects' incomplete four-image blocks of an 8 x 8-output layer would open one more round of blocks, the last n mod 4 images of
 *                        every object go to ONE grouped wave-split-K launch instead (answers differ from "0" by fp32 summation order on those images)
 *   "multi_mid_group" (1): aae_encode_nn_multi -- objects with 5 or more detections share one Winograd launch per conv layer when together they fill
 *                        the chip; 0 = such objects one after the other
 *   [experiments] "winograd" = 2: one launch per polyphase component, the components adding up in the output buffer (2 % slower than the
 *                        one-launch form); "winograd_wide" (0): blocks of 4 waves over both 32-channel halves, one wave per SIMD (13 % slower)
 * small and mid batches (wave-split-K implicit GEMM, conv_wavek_f32.h; the planner: aae_encoder_plan.h)
 *   "wavek" (1), "wavek_dense" (1), "wavek_max_tiles" (512), "wavek_tiny_max_tiles" (64), "wavek_narrow_max_tiles" (128): which layers
 *                        run it and the wave-tile thresholds of the per-detection batches (B = 1, 2, 4)
 *   "wavek_target_blocks" (0 = the device's compute-unit count): blocks of one round of the chip
 *   "wavek_balance" (1), "planner_cost_model" (1), "planner_cost_min_batch" (5), "planner_cost_batch3" (1), "wavek_eff64x32_pct" (74),
 *   "wavek_g_boost" (2), "wavek_tail_split" (1): the planner by estimated time (B = 3, 5 <= B < 256) and its constants
 *   "wavek_force_tail_tiles" (0), "wavek_force_tail_g" (2): tests -- cut the last n tiles of every un-split layer g ways
 *   "dense_gemv_max_batch" (8): the dense layer as a weight-streaming GEMV up to this batch
 *   "ticket_prep" (1): conv1 installs the ticket nonces of the later launches of its call (0: every launch installs its own)
 *   "multi_force_shape" (0), "multi_force_g" (0): A/B of the grouped query's plan (tools/multi_plan_ab.py): wave tile (one nibble per conv
 *                        layer: 1 = 32 x 32, 2 = 64 x 32, 3 = 64 x 64) / K split (one byte per conv layer) forced
 *   [experiments] "wavek_waves" (4 | 8), "wavek_tiny_waves" (4 | 8), "wavek_depth" (2 | 3), "wavek_pingpong" (0), "wavek_spread" (3):
 *                        eight waves per block, three slabs in flight, the barrier-paced schedule, the burst load schedules -- each
 *                        measured slower than the default beside it (CHANGELOG.md rounds 2-4)
 *   [experiments] "gemv_ticket" (1): 0 = the dense GEMV's chunk rows added by a second launch
 *   [experiments] "detect_chain" (0), "detect_chain_blocks" (256): B <= 4 as conv1 + ONE persistent launch with grid barriers --
 *                        bit-identical, 92 vs 82 us at B = 1
 *   [experiments] "wavek_timeline", "chain_timeline": in-kernel phase stamps (aae_encoder_debug_timeline)
 *   [experiments] "wavek_ablate": timing experiments that switch parts of the K loop off -- RESULTS ARE WRONG while != 0
 */

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling aid of the experiments build (tools/ablate_wavek.py, tools/chain_timeline.py): with "wavek_timeline" = 1 wave 0 of every
 * block of the small-batch igemm stamps the shader clock at 8 phase boundaries; this copies the [3 layers][512 blocks][8] stamps of the
 * most recent forward to host memory (synchronises the device).  The product build answers AAE_ERR_INVALID (the option is off). */
int aae_encoder_debug_timeline(aae_encoder* enc, long long* host_out);

#ifdef __cplusplus
}
#endif
#endif /* AAE_HIP_TUNING_H_ */
