/* aae_hip.h -- C ABI of libaae_hip.so: the MI355X (gfx950) implementation of the
 * AugmentedAutoencoder orientation-inference hot path.
 *
 * The reference has no FFI for this path: its boundary is a Python class API
 * whose back end is tf.Session.run.  Each entry point below
 * names the reference interface it replaces (paths relative to /root/reference).
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; device buffers are raw device pointers owned by the
 *     caller (e.g. torch tensors); `stream` is a hipStream_t passed as void*.
 *   - every call returns AAE_OK (0) or a negative error code; the message for
 *     the calling thread is available from aae_last_error().  Nothing exits
 *     the process (the reference print()+exit(-1)s on fatal errors).
 *   - handles own their device weights / codebook and are immutable after
 *     creation (aae_codebook_update excepted); concurrent calls on distinct
 *     streams with distinct workspaces are allowed.  No allocation happens
 *     inside forward / nn / similarity: scratch comes from the caller-provided
 *     workspace sized by the *_workspace_bytes queries.
 */
#ifndef AAE_HIP_H_
#define AAE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AAE_ABI_VERSION 3

#define AAE_OK 0
#define AAE_ERR_INVALID (-1)      /* bad argument                                  */
#define AAE_ERR_UNSUPPORTED (-2)  /* shape / dtype outside what the kernels cover  */
#define AAE_ERR_RUNTIME (-3)      /* HIP runtime error (message has the detail)    */
#define AAE_ERR_WORKSPACE (-4)    /* workspace too small / misaligned              */

#define AAE_DTYPE_U8 0            /* uint8 crops: converted as float32(v/255.)     */
#define AAE_DTYPE_F32 1
#define AAE_DTYPE_BF16 2          /* codebook storage only: bfloat16 rows (2 B/elem), J == 128 */

#define AAE_MAX_LAYERS 8

#define AAE_SCAN_AUTO 0           /* B <= 4: streaming kernel that answers inside its launch; B > 4, top-1 / top-2..8, stride 1:
                                    * query-resident MFMA kernel; else the tile-resident MFMA kernels                        */
#define AAE_SCAN_AUTO_PACKED 7    /* AUTO, but the top-1 query-resident scan (B > 4) reads queries normalised and packed by a
                                     launch in front instead of normalising the raw codes in its own prologue: A/B, same answers
                                     (the other A/B modes: include/aae_hip_tuning.h) */

typedef struct aae_encoder aae_encoder;
typedef struct aae_codebook aae_codebook;
typedef struct aae_decoder aae_decoder;

/* Shapes of Encoder(input, latent_space_size, num_filters, kernel_size, strides,
 * batch_norm)  -- auto_pose/ae/encoder.py:14, filled from the [Network]/[Dataset]
 * cfg keys exactly as ae_factory.build_encoder does (auto_pose/ae/ae_factory.py:33-48). */
typedef struct aae_encoder_desc {
    int32_t in_h, in_w, in_c;             /* [Dataset] H, W, C                      */
    int32_t num_layers;
    int32_t num_filters[AAE_MAX_LAYERS];  /* [Network] NUM_FILTER                   */
    int32_t strides[AAE_MAX_LAYERS];      /* [Network] STRIDES                      */
    int32_t kernel_size;                  /* [Network] KERNEL_SIZE_ENCODER          */
    int32_t latent_size;                  /* [Network] LATENT_SPACE_SIZE            */
    int32_t batch_norm;                   /* [Network] BATCH_NORMALIZATION (0/1)    */
    float bn_eps;                         /* tf.layers.batch_normalization eps 1e-3 */
} aae_encoder_desc;

int aae_abi_version(void);
/* 1 in the experiments build (-DAAE_EXPERIMENTS: every kernel variant that measured slower than the defaults, the profiling and
 * ablation aids -- tools/ and the A/B tests), 0 in the product library (include/aae_hip_tuning.h lists what lives where) */
int aae_has_experiments(void);
const char* aae_last_error(void);

/* ---- Encoder: auto_pose/ae/encoder.py:37-68 (encoder_out + z) -----------------
 * host_weights (float32, host memory), in TF variable order:
 *   per conv layer i : kernel HWIO [k,k,Cin,Cout], bias [Cout]
 *                      (+ gamma, beta, moving_mean, moving_variance [Cout] if batch_norm)
 *   then             : dense kernel [Ho*Wo*C_last, latent], dense bias [latent]
 * Replaces graph construction + factory.restore_checkpoint (ae_factory.py:149-172). */
int aae_encoder_create(const aae_encoder_desc* desc, const void* const* host_weights, int n_weights,
                       aae_encoder** out);
void aae_encoder_destroy(aae_encoder* enc);

/* Options (set before sizing the workspace).  The ones a caller may want:
 *   "precision" (0): 0 = exact fp32 matrix-core arithmetic (bitwise an fp32 fma chain);
 *                    1 = "f32x3h": fp32 in/out, every product of conv2..dense evaluated as three fp16 MFMAs on (hi, lo) operand
 *                        pairs with fp32 accumulation (>= 22-bit operands).  Explicit opt-in; same parity tolerances; activations
 *                        then live in the workspace as fp16 (hi, lo) pairs of x * 2^x3h_act_shift, interleaved per 32-channel chunk.
 *                    2 = f32x3h where it is faster: batches whose first implicit-GEMM layer has at least 256 64x64 output tiles
 *                        (B >= 4 of the default net) run as 1, smaller ones as 0 -- per-detection batches are faster AND exact on
 *                        the fp32 wave-split-K path.
 *   "x3h_act_shift" (4): power-of-two activation pre-scale of the f32x3h format (|x| < 4094 keeps full accuracy; larger values
 *                        saturate gracefully up to 2x and raise the range flag, aae_encoder_x3h_poll).
 *   "compact_workspace" (0): two alternating activation buffers instead of one per layer (805 instead of 973 MB at B = 256; layer
 *                        outputs are then not inspectable through aae_encoder_activation_info).
 *   "dense_gemv" (1): batches of <= 8 run the dense layer as a weight-streaming GEMV instead of a padded matrix-core tile
 *                        (same value up to fp32 summation order).
 *   "multi_group_plan" (1): aae_encode_nn_multi -- a group of objects runs ONE launch plan chosen for the group's total tile count
 *                        (answers differ from per-object aae_encode_nn calls by fp32 summation order only); 0 = every object its
 *                        own plan: bit-identical to aae_encode_nn.
 * Everything else the call accepts -- the planner's constants, kernel-variant switches for A/B measurements -- is listed in
 * include/aae_hip_tuning.h.  In this (product) build every accepted value gives results that are bit-identical to the defaults
 * or differ by fp32 rounding / summation order only; variants that measured slower and the profiling aids live in the experiments build
 * (aae_has_experiments()) and are refused here with AAE_ERR_UNSUPPORTED. */
int aae_encoder_set_option(aae_encoder* enc, const char* name, int value);

size_t aae_encoder_workspace_bytes(const aae_encoder* enc, int B);

/* z = Encoder.z for a batch of crops; replaces session.run(encoder.z, {encoder.x: x})
 * (auto_pose/ae/codebook.py:145,206) and the x/255. of codebook.py:58-59.
 * x: device [B,H,W,C] NHWC (BGR), uint8 or float32.  z_out: device [B, latent] float32. */
int aae_encoder_forward(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out,
                        void* workspace, size_t ws_bytes, void* stream);

/* Same, with HIP events around every kernel launch on `stream` (synchronises the
 * stream).  kernel_ms[i] receives the duration of launch i; returns the launch
 * count through n_kernels.  Labels / algorithmic FLOPs of launch i for this B
 * come from the two queries below (valid after any forward with the same B). */
int aae_encoder_forward_timed(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out,
                              void* workspace, size_t ws_bytes, void* stream,
                              float* kernel_ms, int max_kernels, int* n_kernels);
const char* aae_encoder_kernel_label(const aae_encoder* enc, int i);
double aae_encoder_kernel_flops(const aae_encoder* enc, int i);

/* Byte offset / element count of layer `layer`'s activation [B,Ho,Wo,Cout] inside the
 * workspace after a forward with batch B (parity tests of encoder.py:41-54 per layer). */
/* f32x3h range check ("precision" = 1 only).  Activations travel between layers as fp16 (hi, lo) pairs of
 * x * 2^x3h_act_shift; a pair carries |x * 2^shift| < 65504 (default shift 4: |x| < 4094) at full accuracy.  Every
 * kernel that writes pairs raises a sticky device flag when a value falls outside; this call waits for `stream`,
 * returns the flag and clears it.  1 = the latents of the forwards since the last call may be inaccurate: recompute them
 * with "precision" = 0 (the Python mirror does that automatically).  Exact fp32 mode never sets the flag. */
int aae_encoder_x3h_saturated(aae_encoder* enc, int* flag_out, void* stream);

/* The same check without a wait per forward.  Every f32x3h forward raises its OWN flag: one of 256 slots taken round-robin
 * (a forward recorded into a HIP graph gets one of 64 slots that are never reused -- the graph bakes the address).
 * aae_encoder_x3h_last_slot(): slot of the most recent forward / aae_encode_nn issued by the calling thread, -1 when that
 * forward ran in exact fp32.  aae_encoder_x3h_poll(): waits for `stream` ONCE, returns the flags of `n` slots (1 = that
 * forward met an out-of-range activation: recompute it with "precision" = 0) and clears the raised ones.  A caller can so
 * queue any number of forwards without a host round trip and check them when it consumes the results (the Python mirror:
 * EncoderEngine.settle()).  With more than 256 un-polled forwards a slot is shared: a raised flag may then belong to either
 * user -- a spurious fp32 recomputation at worst, never a missed one.  aae_encoder_x3h_saturated() reports and clears all
 * slots at once. */
int aae_encoder_x3h_last_slot(void);
int aae_encoder_x3h_poll(aae_encoder* enc, const int* slots, int n, int* flags_out, void* stream);
/* A forward recorded into a HIP graph owns its slot (one of 64) until the owner of the graph gives it back: call this when
 * the graph is destroyed (the Python mirror: CapturedNearestNeighbour.close()).  Slots of eager forwards need no release. */
int aae_encoder_x3h_release_slot(aae_encoder* enc, int slot, void* stream);   /* the flag is reset asynchronously on `stream` */

/* 1 when a forward of batch B on this handle runs in f32x3h (precision 1, or precision 2 and a large enough batch):
 * the layer outputs in the workspace are then fp16 (hi, lo) pairs instead of fp32 (aae_encoder_activation_info). */
int aae_encoder_split_precision_for_batch(const aae_encoder* enc, int B);

int aae_encoder_activation_info(const aae_encoder* enc, int B, int layer, size_t* offset_bytes,
                                size_t* count);

/* ---- Codebook: auto_pose/ae/codebook.py:18-51 ---------------------------------
 * E: [N, J] rows already normalised (embedding_normalized variable, codebook.py:28-36, as left
 * by update_embedding :214-216); float32 (AAE_DTYPE_F32, the reference's storage) or bfloat16
 * bit patterns (AAE_DTYPE_BF16, J == 128: half the HBM bytes per scan; queries keep fp32
 * accuracy as three bf16 terms).  Host or device source. */
int aae_codebook_create(const void* E, int N, int J, int dtype, int src_is_device, aae_codebook** out);
int aae_codebook_update(aae_codebook* cb, const void* E, int src_is_device, void* stream);  /* embedding_assign_op */
void aae_codebook_destroy(aae_codebook* cb);
int aae_codebook_set_scan_mode(aae_codebook* cb, int mode);     /* AAE_SCAN_AUTO (default) | AAE_SCAN_AUTO_PACKED | the A/B modes of aae_hip_tuning.h */
/* Upright search (codebook.py:65-66: arg-max over columns 0, k, 2k, ... of the similarity, k = num_cyclo): builds /
 * refreshes a compacted device copy of every col_stride-th row, so that aae_codebook_nn(col_stride = k) scans N/k
 * rows instead of masking a full scan (same scores, same tie rule).  Allocates; call it once outside timed /
 * graph-captured regions.  aae_codebook_update keeps the copy in step.  Without it nn falls back to the masked scan. */
int aae_codebook_prepare_upright(aae_codebook* cb, int col_stride, void* stream);

size_t aae_codebook_workspace_bytes(const aae_codebook* cb, int B, int topk);

/* Nearest rotation indices for un-normalised latents z [B,J] (device):
 * l2_normalize (codebook.py:27) -> matmul transpose_b (:50) -> argmax (:64-68) or
 * top-n (:69-71).  col_stride = 1, or num_cyclo for `upright` (:66; only with topk 1).
 * idx_out: device int64 [B, topk]; score_out: device float32 [B, topk] (cosine).
 * Ties resolve to the lowest index (np.argmax); top-k is score-descending,
 * index-ascending among equal scores.  topk 2..8 with B > 4 is computed inside the scan
 * (sorted per-lane lists); other top-k requests materialise the [B,N] similarity in the
 * workspace first -- aae_codebook_workspace_bytes(cb, B, topk) accounts for either. */
int aae_codebook_nn(aae_codebook* cb, const float* z, int B, int topk, int col_stride,
                    int64_t* idx_out, float* score_out, void* workspace, size_t ws_bytes, void* stream);

/* The same query `reps` times back to back, queued from C between two HIP events on `stream`: *period_ms = device time per
 * query = its kernel(s) + the dependent-launch gap, without the caller's per-call host cost (a Python loop spends longer per
 * call than the 12 us kernel of the B = 1 query takes).  bench.py's `scan.kernel_period_us`, measured in the run that
 * reports it.  (One call between two events on an idle stream measures the events' own latency more than the kernel.)
 * Waits for the second event. */
int aae_codebook_nn_timed(aae_codebook* cb, const float* z, int B, int topk, int col_stride,
                          int64_t* idx_out, float* score_out, void* workspace, size_t ws_bytes, void* stream, int reps, float* period_ms);

/* aae_encoder_forward followed by aae_codebook_nn(topk = 1) on `stream` in ONE call: what
 * Codebook.nearest_rotation(session, x) does per detection (auto_pose/ae/codebook.py:55-68,
 * m3_interface/ae_pose_estimator.py:143-170).  Same results as the two calls, bit for bit.  For B <= 4 the query is
 * six launches: the encoder's first kernel prepares the ticket words of the later launches (wave-split-K tiles, dense
 * GEMV chunks, scan block partials), each of which then finishes its own reduction ("last block to arrive").
 * z_out: device [B, latent] (also returned: Encoder.z).  Workspaces as for the two separate calls. */
int aae_encode_nn(aae_encoder* enc, aae_codebook* cb, const void* x, int x_dtype, int B, int col_stride, float* z_out,
                  int64_t* idx_out, float* score_out, void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace,
                  size_t cb_ws_bytes, void* stream);

/* Full cosine-similarity matrix cs_out [B,N] (device) = session.run(cos_similarity)
 * (codebook.py:63); parity / debugging path, the nn call never materialises it for topk 1. */
int aae_codebook_similarity(aae_codebook* cb, const float* z, int B, float* cs_out,
                            void* workspace, size_t ws_bytes, void* stream);

/* normalized_embedding_query (codebook.py:27) for test_embedding(normalized=True) (:135-145). */
int aae_l2_normalize(const float* z, int B, int J, float* q_out, void* stream);

/* Multi-GPU gather payload (object-sharded inference, one process per GPU): row pos[i] (pos == NULL: row i) of the
 * fixed-capacity buffer packed[capacity][2] receives (idx[i*stride], float bits of score[i*stride]).  One launch
 * instead of four framework element-wise ops in the step that ends in the RCCL all_gather; rows that belong to other
 * ranks keep the -1 sentinel.  Replaces the per-detection bookkeeping of m3_interface/ae_pose_estimator.py:143-170. */
int aae_pack_pairs(const int64_t* idx, const float* score, const int32_t* pos, int n, int stride, int64_t* packed,
                   void* stream);

/* The way back after the all_gather: gathered[world * rows_per_rank][2] holds every rank's packed buffer; answer i of the
 * batch (i < n <= rows_per_rank) is row i of block owner[i] (owner == NULL: block 0).  Writes idx_out[i] (int64) and
 * score_out[i] (float32) in one launch (replaces a fancy-index gather + two element-wise conversions per step). */
int aae_unpack_pairs(const int64_t* gathered, const int32_t* owner, int n, int rows_per_rank, int64_t* idx_out, float* score_out,
                     void* stream);

/* ---- Caller side ("next" row N1): detector crops for a whole image in one launch -------------
 * AePoseEstimator.extract_square_patch(black_borders=True) + cv2.resize(INTER_LINEAR)
 * (auto_pose/m3_interface/ae_pose_estimator.py:106-131,157-162).
 * img: device uint8 [H,W,C]; boxes: device int32 [D,5] = x, y, w, h, size with
 * size = int(max(h, w) * pad_factor); out: device uint8 [D,out_h,out_w,C]. */
int aae_crop_resize_u8(const void* img, int H, int W, int C, const int32_t* boxes, int D,
                       int out_h, int out_w, void* out, void* stream);

/* All detections of ONE object class of a frame in one call: aae_crop_resize_u8 into `crops` (device scratch
 * [n, in_h, in_w, in_c] uint8, the encoder's input shape) followed by aae_encode_nn on them -- what
 * AePoseEstimator.process does per detected box (m3_interface/ae_pose_estimator.py:143-170: extract_square_patch,
 * cv2.resize, one session.run, np.argmax), for n boxes, without a host round trip per launch.  img / boxes / idx_out /
 * score_out may be device memory or device-accessible pinned host memory (a result written straight into pinned memory
 * needs no copy back, an image read from it no upload).  Same results as the two calls. */
int aae_detect_nn(aae_encoder* enc, aae_codebook* cb, const void* img, int H, int W, int C, const int32_t* boxes, int n,
                  int col_stride, void* crops, float* z_out, int64_t* idx_out, float* score_out,
                  void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace, size_t cb_ws_bytes, void* stream);

/* ---- A frame's detections of SEVERAL objects in one call: one launch per LAYER across the objects ---------------------------
 * The reference keeps one AAE (encoder weights + codebook) per object class in one process -- 30 for T-LESS
 * (auto_pose/cfg_m3vision/m3_config_tless.cfg:10-39, m3_interface/ae_pose_estimator.py:61-78) -- and runs one session.run per
 * detected box (:143-170), whatever class it belongs to.  An item = (that class's encoder, its codebook, the n boxes of the
 * class in this frame, col_stride as in aae_codebook_nn).  Inputs and outputs are concatenated in item order: x / crops
 * [rows,H,W,C], z_out [rows,latent], idx_out [rows] int64, score_out [rows], rows = sum of n (aae_multi_rows).
 * Items with n <= 4 whose per-object call would run the per-detection chain (fp32, default options, fp32 codebook, stride 1
 * or a prepared upright copy) are GROUPED: conv1, every later conv layer, the dense GEMV and the codebook scan each run as ONE
 * launch over all grouped items -- objects with different n included -- every block the per-object launch's block, tickets per
 * (object, tile): a frame with C classes costs 6 launches instead of 6 C; a conv layer whose blocks fill the chip over ALL
 * grouped objects (conv2 from ~9 boxes per frame) runs as one polyphase-Winograd launch instead ("multi_group_winograd").
 * Items with n >= 5 whose conv layers are all eligible for polyphase Winograd (default options) form MID-BATCH groups: one Winograd
 * launch per conv layer across the objects for every layer the group's blocks fill the chip on (eight buckets of ~32 crops fill it
 * like one batch of 256; four classes with six boxes each fill conv2 and conv3 -- the other layers run per object), conv1 and the
 * dense layer likewise, the codebook scans in one launch per row-part count + one reduce launch;
 * where the incomplete four-image blocks of an 8 x 8-output layer would open one more round of blocks, the last n mod 4 images
 * of every object are computed by one grouped launch of the direct kernel ("multi_mid_ragged").
 * A class with a few boxes beyond four (5 ... 8; up to 12 for a frame's only such class) is answered as items of <= 4 boxes inside the
 * per-detection group ("multi_split_items").  All other items are answered by aae_encode_nn inside the same call.
 * Results against one aae_encode_nn call per item: with the defaults a group runs ONE launch plan chosen for the group
 * ("multi_group_plan" = 1) resp. the Winograd form where the object alone would take the direct kernels, so latents differ by
 * fp32 summation order / the two forms' rounding (tests bound it at 5e-6 of the latent scale, measured <= 2.3e-6; indices equal
 * wherever the top-2 cosine gap exceeds that), as a batch of another size does; every answer is checked against the object's own fp64 oracle.  With encoder
 * options "multi_group_plan" = 0 and "multi_mid_group" = 0 the call is bit-identical to the per-object calls.
 * All encoders must share the crop shape and the latent size.  The workspace (aae_multi_workspace_bytes; scan_only = 1 for
 * aae_codebook_nn_multi) holds a slice per grouped item: nothing in it outlives the call. */
typedef struct aae_multi_item {
    aae_encoder* enc;        /* may be NULL for aae_codebook_nn_multi */
    aae_codebook* cb;
    int32_t n;               /* detections of this object (>= 1) */
    int32_t col_stride;      /* 1, or num_cyclo for the upright search */
} aae_multi_item;

size_t aae_multi_workspace_bytes(const aae_multi_item* items, int n_items, int scan_only);
int aae_multi_rows(const aae_multi_item* items, int n_items);
int aae_encode_nn_multi(const aae_multi_item* items, int n_items, const void* x, int x_dtype, float* z_out, int64_t* idx_out,
                        float* score_out, void* workspace, size_t ws_bytes, void* stream);
/* the codebook stage alone: z [rows,J] raw latent codes in item order; grouped items: ONE scan launch over their codebooks */
int aae_codebook_nn_multi(const aae_multi_item* items, int n_items, const float* z, int64_t* idx_out, float* score_out,
                          void* workspace, size_t ws_bytes, void* stream);
/* aae_crop_resize_u8 of all boxes (device or pinned int32 [rows,5], item order) into `crops` (device scratch [rows,H,W,C] uint8)
 * + aae_encode_nn_multi: all launches of a frame in ONE call (what AePoseEstimator.process does box by box). */
int aae_detect_nn_multi(const aae_multi_item* items, int n_items, const void* img, int H, int W, int C, const int32_t* boxes,
                        void* crops, float* z_out, int64_t* idx_out, float* score_out, void* workspace, size_t ws_bytes, void* stream);
/* kernel launches the grouped part of the calling thread's last multi call queued (diagnostics / bench) */
int aae_multi_last_launches(void);

/* ---- Decoder ("next" row N4): auto_pose/ae/decoder.py:36-84 (Decoder.x), inference only ---------
 * Decoder(reconstruction_target, latent_code, num_filters, kernel_size, strides, ...) as
 * ae_factory.build_decoder fills it (auto_pose/ae/ae_factory.py:50-70): num_filters / strides
 * are the REVERSED [Network] NUM_FILTER / STRIDES.
 *   dense(latent -> h0*w0*F0, relu)[+BN] -> reshape [h0,w0,F0]
 *   for F_i, size_i: resize_nearest_neighbor(size_i) -> conv2d(F_i, k, 'same', relu)[+BN]
 *   resize_nearest_neighbor([H,W]) -> conv2d(C, k, 'same', sigmoid)
 * with size_i = [H / prod(strides[i:]), W / prod(strides[i:])] (decoder.py:41).
 * host_weights (float32, host), TF variable order:
 *   dense kernel [latent, h0*w0*F0], bias (+ gamma, beta, moving_mean, moving_variance if BN)
 *   per hidden conv i = 1..L-1: kernel HWIO [k,k,F_{i-1},F_i], bias (+ 4 BN arrays)
 *   output conv: kernel [k,k,F_{L-1},C], bias.
 * The auxiliary mask head (decoder.py:66-73) does not feed Decoder.x and is not evaluated. */
typedef struct aae_decoder_desc {
    int32_t out_h, out_w, out_c;          /* reconstruction target shape = [Dataset] H, W, C */
    int32_t num_layers;                   /* len(NUM_FILTER)                                 */
    int32_t num_filters[AAE_MAX_LAYERS];  /* reversed [Network] NUM_FILTER                   */
    int32_t strides[AAE_MAX_LAYERS];      /* reversed [Network] STRIDES                      */
    int32_t kernel_size;                  /* [Network] KERNEL_SIZE_DECODER                   */
    int32_t latent_size;
    int32_t batch_norm;
    float bn_eps;
} aae_decoder_desc;

int aae_decoder_create(const aae_decoder_desc* desc, const void* const* host_weights, int n_weights,
                       aae_decoder** out);
void aae_decoder_destroy(aae_decoder* dec);
size_t aae_decoder_workspace_bytes(const aae_decoder* dec, int B);

/* x = Decoder.x for a batch of latent codes; replaces session.run(decoder.x, {encoder.x: ...})
 * after the encoder (auto_pose/eval/eval_plots.py:33,59) and
 * session.run(decoder.x, {decoder._latent_code: z}) (eval_plots.py:78).
 * z: device [B, latent] float32 (NOT normalised).  x_out: device [B,H,W,C] float32 in [0,1]. */
int aae_decoder_forward(aae_decoder* dec, const float* z, int B, float* x_out, void* workspace,
                        size_t ws_bytes, void* stream);
int aae_decoder_forward_timed(aae_decoder* dec, const float* z, int B, float* x_out, void* workspace,
                              size_t ws_bytes, void* stream, float* kernel_ms, int max_kernels, int* n_kernels);
const char* aae_decoder_kernel_label(const aae_decoder* dec, int i);
double aae_decoder_kernel_flops(const aae_decoder* dec, int i);
/* workspace region of hidden activation `stage` (0 = dense output, i = i-th hidden conv) */
int aae_decoder_activation_info(const aae_decoder* dec, int B, int stage, size_t* offset_bytes, size_t* count);

#ifdef __cplusplus
}
#endif
#endif /* AAE_HIP_H_ */
