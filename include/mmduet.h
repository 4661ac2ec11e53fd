/*
 * mmduet.h -- C ABI of libmmduet_hip.so: the MI355X-native (gfx950) streaming video-text-duet forward path.
 *
 * The reference (yellow-binary-tree/MMDuet) has no native/FFI interface of its own: its drop-in boundary is the
 * Python duck-type the stream driver uses.  This header is the native layer underneath that
 * Python surface; every entry point names the reference interface it stands in for (paths relative to the reference
 * repository root).  The Python shim that binds it with ctypes is mmduet_amd/_lib.py.
 *
 * Conventions
 *   - all functions return 0 on success or a negative MMD_E* code; mmd_last_error() gives the message.
 *     No C++ exception crosses the ABI.
 *   - pointers are DEVICE pointers unless the parameter is documented as host.
 *   - one hipStream_t per context (mmd_set_stream); calls on one context are not thread-safe -- the Python layer
 *     holds a per-model mutex (the reference's Gradio demo calls the model from two threads, demo/app.py:84-85).
 *   - activation / weight element type is chosen per context (mmd_config.dtype): MMD_BF16 (production) or MMD_F32
 *     (bit-for-bit-class parity runs against the fp32 oracle).  Logit outputs are always fp32, like the reference's
 *     `.float()` (models/live_llava/video_head_live_llava_qwen.py:155,160-161).
 */
#ifndef MMDUET_H
#define MMDUET_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMD_OK 0
#define MMD_EINVAL (-22)
#define MMD_ENOMEM (-12)
#define MMD_ENOENT (-2)
#define MMD_ERANGE (-34)
#define MMD_EHIP (-5)

typedef enum { MMD_F32 = 0, MMD_BF16 = 1 } mmd_dtype;
/* storage of the decoder's linear-layer weights (q/k/v/o/gate/up/down): MMD_W_FP8_E4M3 = OCP e4m3fn with one fp32 scale per output
 * channel (BASELINE configs[4], scripts/inference/youcook2.sh:12-14 workload), activations and accumulation unchanged (bf16 / fp32) */
typedef enum { MMD_W_DTYPE = 0, MMD_W_FP8_E4M3 = 1 } mmd_weight_dtype;
/* ADAPTIVE_AVG = adaptive_avg_pool2d to pool_stride x pool_stride tokens (the secondary encoder path, models/vision_live.py:17-24) */
typedef enum { MMD_POOL_BILINEAR = 0, MMD_POOL_AVERAGE = 1, MMD_POOL_MAX = 2, MMD_POOL_ADAPTIVE_AVG = 3 } mmd_pool_mode;

/* Shape of the model.  Mirrors the fields of VideoHeadLiveLlavaQwenConfig (models/live_llava/video_head_live_llava_qwen.py:41-45,
 * models/configuration_live.py:22-36) plus the SigLIP tower shape LLaVA-NeXT hard-codes. */
typedef struct mmd_config {
    int32_t struct_size;          /* = sizeof(mmd_config), ABI guard */
    int32_t dtype;                /* mmd_dtype of weights and activations */
    /* Qwen2 decoder */
    int32_t vocab_size, hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads, head_dim;
    float rope_theta, rms_norm_eps;
    /* SigLIP tower (vit_layers = layers that RUN) */
    int32_t vit_hidden, vit_intermediate, vit_layers, vit_heads, vit_image, vit_patch;
    float vit_ln_eps;
    int32_t vit_post_layernorm;
    /* connector / pooling (models/live_llava/video_head_live_llava_qwen.py:100-119) */
    int32_t pool_mode, pool_stride, frame_num_tokens;
    /* workspace sizing */
    int32_t max_vit_batch;        /* frames per mmd_vit_encode call (reference: 32, test/inference.py:208) */
    int32_t max_step_tokens;      /* max rows of one mmd_llm_step call */
    int32_t weight_dtype;         /* mmd_weight_dtype (bf16 contexts only) */
    /* secondary frame encoders (models/vision_live.py:11-64): a context may carry ONLY a tower -- HF SigLIP-L/16-384 or CLIP-L/14-336 */
    int32_t vision_only;          /* 1: no decoder / projector; the mmd_vision_* entry points below are the product */
    int32_t vit_class_token;      /* CLIP: class embedding prepended (tokens = grid^2 + 1) */
    int32_t vit_pre_layernorm;    /* CLIP: pre_layrnorm after the embeddings */
    int32_t vit_act;              /* 0 gelu_pytorch_tanh (SigLIP), 1 quick_gelu (CLIP) */
    int32_t vit_pool_head;        /* SigLIP: SiglipMultiheadAttentionPoolingHead present (pooler_output, used when frame_token_cls) */
    int32_t tower_f16;            /* 1: the LLaVA SigLIP tower as under torch.cuda.amp.autocast() (models/modeling_live.py:28): fp16 linears / attention matmuls
                                     with fp32 accumulate, LayerNorm / softmax in fp32, and the hidden state between them in FP32 (`fp32 + fp16` promotes), bf16
                                     features to the projector.  2: the same with the residual stream rounded to fp16 after every sublayer (the round-3 form,
                                     ~2 % faster tower).  bf16 contexts only; 0 = tower in the context dtype */
} mmd_config;

typedef struct mmd_ctx mmd_ctx;
typedef struct mmd_stream mmd_stream;

/* ---- context ------------------------------------------------------------------------------------------------ */
/* replaces model construction in build_live (models/modeling_live.py:80-129) */
int mmd_create(const mmd_config* cfg, int device, mmd_ctx** out);
void mmd_destroy(mmd_ctx* ctx);
const char* mmd_last_error(const mmd_ctx* ctx);          /* ctx may be NULL: error of the last failed mmd_create */
int mmd_set_stream(mmd_ctx* ctx, void* hip_stream);      /* hipStream_t used verbatim: NULL = the null stream (torch's default stream).
                                                            Until called, the context runs on a stream of its own. */
void* mmd_get_stream(mmd_ctx* ctx);
int mmd_synchronize(mmd_ctx* ctx);

/* ---- weights -------------------------------------------------------------------------------------------------- */
/* replaces from_pretrained weight materialisation (models/modeling_live.py:96-99).  `name` is the checkpoint name
 * (e.g. "model.layers.3.self_attn.q_proj.weight"); data is row-major in `src_dtype`; on_device != 0 if `data` is a
 * device pointer.  Tensors are converted to the context dtype.  LoRA deltas (models/modeling_live.py:123, peft
 * y = Wx + (alpha/r) B(Ax)) are merged with mmd_merge_lora before mmd_finalize_weights. */
int mmd_load_tensor(mmd_ctx* ctx, const char* name, const void* data, int src_dtype, const int64_t* shape, int rank,
                    int on_device);
int mmd_merge_lora(mmd_ctx* ctx, const char* weight_name, const float* A_host, const float* B_host, int r, float scale);
/* RoPE inverse frequencies theta^(-2i/d), i < head_dim/2 (host fp32).  Optional: the default table is computed in
 * double precision; the Python shim passes the table torch computes with the reference's own expression
 * (transformers qwen2/modeling_qwen2.py:84-85) so that fp32 parity at large positions does not hinge on pow() ulps. */
int mmd_set_rope_inv_freq(mmd_ctx* ctx, const float* inv_freq_host, int n);
int mmd_finalize_weights(mmd_ctx* ctx);                  /* builds fused layouts; fails listing the first missing tensor */
int64_t mmd_weight_bytes(const mmd_ctx* ctx);

/* ---- vision side ---------------------------------------------------------------------------------------------- */
/* Scheduling knob, no effect on results: cap the persistent grid of the tower's / projector's ring GEMMs at `max_blocks` workgroups (0 = one per CU).
 * The driver (mmduet_amd/inference.py) halves the tower's share while a response is being decoded on the other HIP stream: the weight-streaming
 * decode kernels then find free CUs and tower batches hide under the decode burst instead of queueing in front of it. */
int mmd_set_tower_share(mmd_ctx* ctx, int max_blocks);
/* replaces LiveMixin.visual_embed (models/modeling_live.py:26-33): tower -> connector -> pooling.
 * pixel_values [B,3,img,img] in ctx dtype; out [B*frame_num_tokens, hidden] in ctx dtype. */
int mmd_vit_encode(mmd_ctx* ctx, const void* pixel_values, int B, void* out);
/* preprocess + visual_embed in one call: uint8 frames [B,3,R,R] (device) -> out [B*frame_num_tokens, hidden].  The last
 * resampler pass writes the normalised pixels straight into the im2col matrix of the patch-embed GEMM; results are bit-identical to
 * mmd_preprocess_frames followed by mmd_vit_encode. */
int mmd_vit_encode_frames(mmd_ctx* ctx, const uint8_t* frames, int B, int R, void* out);
/* second half of visual_embed for pre-extracted tower features (the reference's offline feature files, data/utils.py:99-117: per-video
 * [T, tokens, C] written from `vision_encode`; visual_embed without a tower starts at the connector, models/modeling_live.py:26-33):
 * tower_features [B, tokens, vit_hidden] (ctx dtype) -> mm_projector -> post_projector_pooling -> out [B*frame_num_tokens, hidden] */
int mmd_connector_pool(mmd_ctx* ctx, const void* tower_features, int B, void* out);
/* ---- secondary encoder path (models/vision_live.py:11-54 `_siglip_vision_encode` / `_clip_vision_encode`) -------------------------------
 * mmd_normalize_frames: torchvision normalize(frames * rescale, mean, std) (:13,:36); frames uint8 (src_kind 0) or fp32 (1) [B,3,R,R] -> ctx dtype.
 * mmd_vision_tower: `vision_model(frames).last_hidden_state` -> out [B, tokens, vit_hidden] (SigLIP: after post_layernorm; CLIP: class token first,
 *   pre_layrnorm, quick_gelu, NO post_layernorm on the sequence).
 * mmd_vision_pool_tokens: adaptive_avg_pool2d of the s x s spatial tokens (class token skipped) to out_h x out_w (:17-24,:40-47) -> [B, out_h*out_w, C].
 * mmd_vision_pool_head: SigLIP pooler_output (probe attention + LayerNorm + MLP, siglip/modeling_siglip.py [3P]) -> [B, C] (:28). */
int mmd_normalize_frames(mmd_ctx* ctx, const void* frames, int src_kind, int B, int R, const float* mean3_host, const float* std3_host, float rescale, void* out);
int mmd_vision_tower(mmd_ctx* ctx, const void* pixel_values, int B, void* out);
int mmd_vision_pool_tokens(mmd_ctx* ctx, const void* feats, int B, int out_h, int out_w, void* out);
int mmd_vision_pool_head(mmd_ctx* ctx, const void* feats, int B, void* out);
/* intermediate taps for parity tests: stage 0 = tower output [B*tokens, vit_hidden], 1 = connector output
 * [B*tokens, hidden]; valid until the next mmd_vit_encode. */
/* The default encode path runs the LAST tower layer and the projector on the tokens the bilinear post_projector_pooling reads ((2 out)^2 of vit_tokens per frame:
 * models/live_llava/video_head_live_llava_qwen.py:100-119 interpolates without antialiasing) -- same values, same arithmetic, rows nobody reads are not computed.  Callers
 * that want vision_encode's full [B, tokens, C] (offline feature extraction, data/utils.py:99-117) or the debug taps switch that off first. */
int mmd_vit_set_full_tower(mmd_ctx* ctx, int on);
int mmd_vit_get_full_tower(const mmd_ctx* ctx);            /* current setting (1 / 0): callers that flip it temporarily restore what they found */
int mmd_vit_debug_tap(mmd_ctx* ctx, int stage, void* out, int64_t out_elems);
/* replaces image_processor.preprocess (test/inference.py:203; LLaVA SigLipImageProcessor): uint8 [T,3,R,R] ->
 * PIL-bicubic resize to img x img (bit-exact with Pillow's 8-bit resampler), x/255, (x-.5)/.5 -> ctx dtype. */
int mmd_preprocess_frames(mmd_ctx* ctx, const uint8_t* frames, int T, int R, void* pixel_values);
/* replaces the per-frame resize + pad + colour flip of load_video (test/datasets.py:52-71, demo/liveinfer.py:32-51):
 * decoded frames uint8 [T,H,W,3] (device) -> cv2.resize(INTER_LINEAR, 8-bit fixed point) to the letterbox size ->
 * cv2.copyMakeBorder(BORDER_CONSTANT, pad_color[3], in SOURCE channel order) -> cv2.cvtColor(BGR2RGB) when flip_channels
 * -> CHW, out uint8 [T,3,R,R].  Geometry (target size and the four pad widths) as the reference computes it. */
int mmd_letterbox_geometry(int W, int H, int R, int* new_w, int* new_h, int* top, int* bottom, int* left, int* right);
int mmd_letterbox_frames(mmd_ctx* ctx, const uint8_t* frames, int T, int H, int W, int R, const uint8_t* pad_color,
                         int flip_channels, uint8_t* out);

/* ---- language side -------------------------------------------------------------------------------------------- */
/* replaces model.get_input_embeddings()(ids) (test/inference.py:236,251,259; models/modeling_live.py:76). ids: device int64 */
int mmd_embed_tokens(mmd_ctx* ctx, const int64_t* ids, int k, void* out);

/* KV arena of one video stream; replaces the DynamicCache / legacy tuple cache handed around as `past_key_values`
 * (test/inference.py:183,239-240; the reference regrows it by torch.cat every step).  O(1) append and O(1) truncate.  The arena is a
 * VIRTUAL address range sized for MMDUET_KV_VIRTUAL_TOKENS (default 4 Mi tokens ~ the 288 GB limit of the 7B model) whose first
 * mmd_kv_capacity tokens are backed by physical pages; growth maps more pages (hipMemCreate / hipMemMap) behind the same addresses --
 * nothing is copied and no second arena ever exists.  (MMDUET_KV_NO_VMM=1: plain allocation, growth by reallocation + copy.) */
int mmd_stream_create(mmd_ctx* ctx, int64_t initial_tokens, mmd_stream** out);
void mmd_stream_destroy(mmd_stream* s);
int64_t mmd_kv_len(const mmd_stream* s);
int64_t mmd_kv_capacity(const mmd_stream* s);             /* tokens backed by memory */
int64_t mmd_kv_stride(const mmd_stream* s);               /* tokens per (layer, kv head) row of the address range */
int mmd_kv_truncate(mmd_stream* s, int64_t new_len);      /* rollback: remove_assistant_turns (test/inference.py:265-269), speculative chunks */
/* set the KV of tokens [from, to) aside / bring it back (length becomes `to`): with remove_assistant_turns (test/inference.py:265-269) a response fired in
 * the middle of a multi-frame chunk overwrites the slots behind it and is then dropped -- the frames already encoded behind it are restored, not recomputed */
int mmd_kv_stash(mmd_stream* s, int64_t from, int64_t to);
int mmd_kv_unstash(mmd_stream* s);
int mmd_stream_reset(mmd_stream* s);                      /* LiveInferForBenchmark.reset (test/inference.py:169-183: past_key_values = None): length 0, arena kept */
int mmd_kv_debug_set_len(mmd_stream* s, int64_t n);       /* measurement aid: mark n slots live without computing them */

/* replaces VideoHeadLiveLlavaQwenForCausalLM.forward body (models/live_llava/video_head_live_llava_qwen.py:141) ==
 * Qwen2Model.forward for batch 1: embeds [S, hidden] are appended at positions kv_len..kv_len+S-1; hidden_out
 * [S, hidden] receives the post-final-RMSNorm hidden state (may be NULL). */
int mmd_llm_step(mmd_ctx* ctx, mmd_stream* s, const void* embeds, int S, void* hidden_out);
/* informative_head / relevance_head (models/live_llava/video_head_live_llava_qwen.py:160-161): hidden rows [M, hidden] ->
 * out [M,4] fp32 = (informative[0], informative[1], relevance[0], relevance[1]). */
int mmd_video_heads(mmd_ctx* ctx, const void* hidden, int M, float* out);
/* lm_head (models/live_llava/video_head_live_llava_qwen.py:155): hidden rows [M, hidden] -> logits [M, vocab] fp32 */
int mmd_lm_head(mmd_ctx* ctx, const void* hidden, int M, float* logits);
/* one fused per-frame step for the streaming loop (test/inference.py:239-244): llm_step + heads at the rows listed in
 * head_rows (host int32[n_rows], e.g. the last token of every frame in a chunk); scores_out host fp32 [n_rows,4]
 * (synchronises the stream). */
int mmd_frame_step(mmd_ctx* ctx, mmd_stream* s, const void* embeds, int S, const int32_t* head_rows_host, int n_rows,
                   float* head_logits_host);

/* Multi-stream forms (no reference counterpart: the reference runs one video per process, batch 1; SURVEY.md section 7 names
 * "batch several independent streams per GPU" as the way out of the weight-streaming regime).  Segment j = seg_rows[j]
 * consecutive rows of `embeds`, extending streams[j] (each stream at most once): the GEMMs run ONCE over all rows -- a stream
 * that is generating rides with its single row on the other streams' frame chunks -- RoPE/KV append/attention run per
 * segment on its own arena.  Per-row results equal the single-stream calls up to GEMM accumulation order.
 * mmd_frame_step_multi adds, after the step: the 4 video-head logits at head_rows -> heads_out_host [n,4] (host, ONE
 * sync when n_head_rows > 0), the final hidden state at hidden_rows -> hidden_rows_out [n,hidden] (device, ctx dtype) and,
 * when logits_out != NULL, lm_head over those rows in one pass -> logits_out [n,vocab] fp32 (device). */
int mmd_llm_step_multi(mmd_ctx* ctx, mmd_stream* const* streams, const int32_t* seg_rows, int n_segs, const void* embeds, void* hidden_out);
int mmd_frame_step_multi(mmd_ctx* ctx, mmd_stream* const* streams, const int32_t* seg_rows, int n_segs, const void* embeds,
                         const int32_t* head_rows, int n_head_rows, float* heads_out_host,
                         const int32_t* hidden_rows, int n_hidden_rows, void* hidden_rows_out, float* logits_out);

/* Scheduler rounds with the greedy sampling on the device (the multi-stream form of fast_greedy_generate, models/modeling_live.py:51-77; the offline drivers
 * test/inference.py:257-274 call it once per response).  With several streams per GPU the token loops of all talking streams advance together, one token per round,
 * inside the forward that also carries the watching streams' frame chunks; per round the host sees the head logits and ONE token id per talking stream -- no logits
 * tensor, no host arg-max, no embedding call.
 * mmd_sampler = greedy sampling state of one stream on the device: the token drawn last and the repetition-penalty list that persists across the turns of a video
 * (models/modeling_live.py:60-66).  mmd_sampler_begin starts a response: eos id, penalty (<= 0: none), the ids generated so far in this video (host), the most tokens
 * this response may add.  The sampler then grows the list itself: a drawn token joins it unless it is EOS (EOS is returned but neither fed back nor penalised).
 * mmd_round_multi = mmd_frame_step_multi with, per segment j, its input rows seg_embeds[j] [seg_rows[j], hidden] (device, ctx dtype) and flags:
 *   MMD_SEG_FEED    the segment is ONE row, the embedding of the token samplers[j] drew in an earlier round (gathered on the device; seg_embeds[j] is ignored);
 *   MMD_SEG_SAMPLE  after the step: lm_head over the segment's last row, repetition penalty over samplers[j]'s list, arg-max (first maximal index) -> tokens_out_host[j]
 *                   and samplers[j]'s token slot.
 * tokens_out_host [n_segs] receives -1 for the other segments; heads as in mmd_frame_step_multi.  ONE stream synchronisation per round (none when nothing is read).
 * At most 32 feed and 32 sampling segments per round. */
typedef struct mmd_sampler mmd_sampler;
#define MMD_SEG_FEED 1
#define MMD_SEG_SAMPLE 2
int mmd_sampler_create(mmd_ctx* ctx, mmd_sampler** out);
void mmd_sampler_destroy(mmd_sampler* sp);                /* before mmd_destroy of its context */
int mmd_sampler_begin(mmd_sampler* sp, int64_t eos_id, float rep_penalty, const int64_t* prev_ids_host, int n_prev, int max_new);
int mmd_sampler_prev_len(const mmd_sampler* sp);          /* entries of the penalty list that count */
int mmd_round_multi(mmd_ctx* ctx, mmd_stream* const* streams, const int32_t* seg_rows, int n_segs, const void* const* seg_embeds, mmd_sampler* const* samplers,
                    const int32_t* seg_flags, const int32_t* head_rows, int n_head_rows, float* heads_out_host, int64_t* tokens_out_host);

/* replaces fast_greedy_generate (models/modeling_live.py:51-77): feeds prompt_embeds [S,hidden], then up to max_new
 * greedy steps.  prev_ids_host / n_prev: the repetition-penalty list persisted across turns (grown in place, capacity
 * prev_cap); rep_penalty <= 0 disables the penalty (and the list is not updated, like the reference).  EOS is written
 * to out_ids but neither fed back nor added to the list.  out_ids_host int64[max_new]; n_out = tokens written. */
int mmd_greedy_generate(mmd_ctx* ctx, mmd_stream* s, const void* prompt_embeds, int S, int64_t eos_id, float rep_penalty,
                        int64_t* prev_ids_host, int* n_prev, int prev_cap, int64_t* out_ids_host, int max_new, int* n_out);

/* ---- multi-GPU: the one collective of the path ------------------------------------------------------------------ */
/* The reference shards videos over N manually launched processes (--start_idx/--end_idx, test/inference.py:337) and has no
 * collective; here one process per GPU gathers the per-frame head scores of its streams with ONE RCCL all-gather over xGMI.
 * Rank 0 draws an id (mmd_comm_unique_id, host bytes), the launcher's rendezvous store hands it to every rank, each rank calls
 * mmd_comm_create (collective).  mmd_gather_scores: local [T,2] fp32 (device; informative, relevance probability per frame)
 * -> all [world, t_max+1, 2] fp32 (device): row 0 of every rank's block = (T, 0), rows 1..T the scores, NaN beyond.  Enqueued
 * on the communicator's stream, no host synchronisation. */
#define MMD_COMM_ID_BYTES 128
typedef struct mmd_comm mmd_comm;
int mmd_comm_unique_id(uint8_t* id_out_host);
int mmd_comm_create(const uint8_t* id_host, int rank, int world, int device, void* hip_stream, mmd_comm** out);
void mmd_comm_destroy(mmd_comm* comm);
int mmd_comm_world(const mmd_comm* comm);
const char* mmd_comm_last_error(const mmd_comm* comm);    /* comm may be NULL: error of the last failed create / unique_id */
int mmd_gather_scores(mmd_comm* comm, const float* local, int T, int t_max, float* all);
int mmd_comm_probe(void);                                          /* MMD_OK if librccl can be bound (asked by every rank before the collective create) */
int mmd_comm_set_stream(mmd_comm* comm, void* hip_stream);         /* the stream the following gathers are issued on */
int mmd_gather_block(mmd_comm* comm, const float* block, int64_t n_floats, float* all);   /* raw: [n_floats] per rank -> [world, n_floats], one ncclAllGather */

/* ---- measurement ---------------------------------------------------------------------------------------------- */
/* HIP-event timing of kernel classes on the context's stream (bench.py `roofline`).  While enabled every launch of a
 * enabled class is bracketed by events; mmd_prof_read returns accumulated ms and launch counts. */
enum { MMD_K_GEMM_SKINNY = 0, MMD_K_GEMM_TILE = 1, MMD_K_ATTN_LLM = 2, MMD_K_ATTN_VIT = 3, MMD_K_NORM_ROPE = 4,
       MMD_K_OTHER = 5, MMD_K_COUNT = 6 };
int mmd_prof_enable(mmd_ctx* ctx, int class_mask);       /* bit k set = bracket launches of class MMD_K_k; 0 = off */
int mmd_prof_set_stride(mmd_ctx* ctx, int stride);       /* bracket only every stride-th launch of an enabled class (sampling) */
int mmd_prof_read(mmd_ctx* ctx, double* ms_out /*[MMD_K_COUNT]*/, int64_t* launches_out /*[MMD_K_COUNT]*/,
                  double* bytes_out /*[MMD_K_COUNT] algorithmic bytes*/, double* flops_out /*[MMD_K_COUNT]*/);
int mmd_prof_reset(mmd_ctx* ctx);

/* ---- raw operator entry points (parity tests call the kernels through these) ------------------------------------ */
/* Y[M,N] = epilogue(X[M,K] . W[N,K]^T + bias).  epi: 0 none, 1 gelu(tanh), 2 gelu(erf), 3 add residual R[M,N],
 * 4 SwiGLU (W rows interleaved gate/up in blocks of 16 -> Y[M,N/2]).  out_f32 != 0 writes fp32. variant: 0 auto,
 * 1 generic tile, 2 skinny/split-K, 3 large tile, 4 DMA 128-row tile, 5 skinny slabs, 6 256x256 ring, 8 streaming kernel (32 < M <= 256; SwiGLU only in this form). */
int mmd_op_gemm(mmd_ctx* ctx, const void* X, const void* W, const void* bias, const void* R, void* Y, int M, int N, int K,
                int epi, int out_f32, int variant);
/* what the dispatcher chose for the most recent GEMM of this context: out4 = {kernel (0 tile64, 1 tile128, 2 skinny, 3 gemv16, 4 big64,
 * 5 big128, 6 ring256), output tiles, K splits, blocks launched} */
int mmd_op_gemm_last_plan(mmd_ctx* ctx, int* out4);
/* a producer GEMM and its ONLY consumer: T = epi1(X[M,K1] . W1[N1,K1]^T + b1), Y[M,N2] = T . W2[N2,T]^T (+ R).  piece_major != 0 lets the intermediate live in the ring
 * kernel's piece-major activation layout when both GEMMs take that kernel (what the tower's fc1 -> fc2 and a chunk's gate_up -> down do inside the model); *used_pm_out says
 * whether it did.  Same bits either way. */
int mmd_op_gemm_pair(mmd_ctx* ctx, const void* X, const void* W1, const void* b1, int epi1, const void* W2, const void* R, void* Y, int M, int N1, int K1, int N2,
                     int piece_major, int* used_pm_out);
/* the weight-streaming GEMMs in slab mode (what the fused LLM schedule launches for M <= 256): `*splits_out` fp32 partial slabs [splits][M][N] of X . W^T in slabs_out
 * (device, room for max_splits slabs); variant 2 = the dispatcher's choice by M (gemv16 / skinny / stream), 8 = gemm_stream_kernel */
int mmd_op_gemm_slabs(mmd_ctx* ctx, const void* X, const void* W, int M, int N, int K, int variant, float* slabs_out, int max_splits, int* splits_out);
/* fp8 weight path, raw form: mmd_op_quantize_fp8 replaces W [N,K] (bf16, row-major, device) by bf16(q), q = rne_e4m3(W / scale[n]),
 * scale[n] = amax_n / 448, and writes the fp8 bytes q8 [N,K] and scale [N] (fp32).  mmd_op_gemm_w8 = mmd_op_gemm on such a matrix:
 * Y = epilogue((X . q^T) * scale + bias); M <= 64 streams the fp8 copy, larger M the bf16(q) copy (bit-identical values). */
int mmd_op_quantize_fp8(mmd_ctx* ctx, void* W, int N, int K, uint8_t* q8_out, float* scale_out);
int mmd_op_gemm_w8(mmd_ctx* ctx, const void* X, const void* Wq, const uint8_t* q8, const float* scale, const void* bias, const void* R, void* Y,
                   int M, int N, int K, int epi, int out_f32, int variant);
/* micro-benchmark of one GEMM shape (HIP events on the context's stream); avg ms per call incl. any split-K reduce.  X [M,K] / W [N,K]
 * (device, ctx dtype) supply the operand VALUES (random data: operand bits set the chip's clock); NULL = a constant fill */
int mmd_op_gemm_bench(mmd_ctx* ctx, int M, int N, int K, int epi, int variant, int iters, float* avg_ms_out, const void* X, const void* W);
int mmd_op_rmsnorm(mmd_ctx* ctx, const void* x, const void* w, void* y, int M, int H, float eps);
int mmd_op_layernorm(mmd_ctx* ctx, const void* x, const void* w, const void* b, void* y, int M, int H, float eps);
/* the autocast tower's residual step (models/modeling_live.py:28: `hidden (fp32) + sublayer_out (fp16)` promotes, LayerNorm returns fp32, the next linear casts to
 * fp16): h32[M,H] = (pos16 ? float(pos16[m % period]) : h32) + float(y16); out16 = fp16(LayerNorm(h32) * w16 + b16) when w16 is given; outbf = bf16(h32) when outbf
 * is given (h32 itself is not rewritten then).  All 16-bit operands are IEEE half except outbf.  H % 8 == 0, H <= 2048. */
int mmd_op_resid32_layernorm(mmd_ctx* ctx, const void* y16, float* h32, const void* pos16, int period, const void* w16, const void* b16, void* out16, void* outbf,
                             int M, int H, float eps);
/* q [S, nh*d] (rotated in place), k/v [S, nkv*d] appended rotated/unrotated at pos0.. into Kc/Vc [nkv, cap, d] */
int mmd_op_rope_append(mmd_ctx* ctx, void* qkv, int S, int nh, int nkv, int d, float theta, int64_t pos0, void* q_out,
                       void* Kc, void* Vc, int64_t cap);
/* causal GQA attention with query offset: q [S, nh*d], Kc/Vc [nkv, cap, d], n_ctx = tokens before this step;
 * out [S, nh*d].  causal = 0 -> full attention over n_ctx + S keys.  variant: 0 auto, 1 simple, 2 mfma, 3 the arena kernels (attn_gqa128_kernel), 4 row-major
 * K / V (the tower's), 5 attn_gqa128_w1_kernel, 6 attn_gqa128_chunk_kernel. */
int mmd_op_attention(mmd_ctx* ctx, const void* q, const void* Kc, const void* Vc, void* out, int S, int nh, int nkv, int d,
                     int64_t n_ctx, int64_t cap, int causal, int variant);
int mmd_op_attention_bench(mmd_ctx* ctx, int S, int nh, int nkv, int d, int64_t n_ctx, int variant, int iters, float* avg_ms_out);
/* which attention form the most recent LLM (or raw-operator) attention launch of THIS context took (what mmd_op_gemm_last_plan is for the GEMMs: parity tests assert that
 * the production kernel really ran; the tower's launches and other contexts do not disturb it): 1 one wave per row, 2 16-row MFMA tiles, 3 / 4 attn_gqa128_kernel with 16- /
 * 32-row waves (decode and two-slot forms / 256-row phase-split chunks), 5 attn_gqa128_w1_kernel (per-frame steps, short chunks), 6 register-staged row-major (ViT), 7
 * attn_d72_ring_kernel (SigLIP-so400m), 8 attn_gqa128_chunk_kernel (multi-frame chunks, contiguous decomposition), 9 the decode rows of several streams in one launch
 * (mmd_round_multi); out2 = {form, key splits or the most blocks sharing a unit} */
int mmd_op_attention_last_form(mmd_ctx* ctx, int* out2);
int mmd_op_pool(mmd_ctx* ctx, const void* x, void* y, int B, int grid, int H, int mode, int stride);

#ifdef __cplusplus
}
#endif
#endif
