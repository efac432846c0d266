/* q3c - C ABI of the MI355X (gfx950) Qwen3-TTS 12 Hz speech-tokenizer decode library.
 *
 * The reference repository (mindvridge/Qwen3-TTS) has no FFI of its own for this path: the decode is
 * Python inside the `qwen-tts` dependency and is reached implicitly at the end of
 * `Qwen3TTSModel.generate_voice_clone` (reference server.py:245-251) and `generate_custom_voice`
 * (reference models.py:96-101).  Each entry point below names the Python-level operation it stands in
 * for; the sibling lines (`sib` = transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py,
 * `mimi` = transformers/models/mimi/modeling_mimi.py) give the arithmetic it must reproduce.
 *
 * Conventions
 *  - plain C: no C++ types, no exceptions.  Every function returns 0 on success or a negative q3c_status;
 *    q3c_last_error(ctx) returns a message owned by the context (valid until the next call on it).
 *  - device pointers are owned by the caller; the library never frees or reallocates them.  The library
 *    owns only what q3c_finalize() allocated (repacked weights, RoPE tables) and frees it in q3c_destroy().
 *  - launch functions take a hipStream_t (passed as void*), are asynchronous and never synchronise.
 *  - one context per (process, device); calls on one context must not overlap; distinct contexts are
 *    independent.
 *  - activations are channels-last: [batch][time][channel].
 */
#ifndef Q3C_H
#define Q3C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3C_MAX_STAGES 8

typedef enum q3c_status {
    Q3C_OK = 0,
    Q3C_ERR_INVALID_ARGUMENT = -1,   /* bad size / pointer / configuration */
    Q3C_ERR_WEIGHT = -2,             /* unknown name, wrong shape, missing tensor */
    Q3C_ERR_STATE = -3,              /* call order (e.g. forward before finalize) */
    Q3C_ERR_WORKSPACE = -4,          /* workspace too small */
    Q3C_ERR_HIP = -5,                /* a HIP runtime call failed */
    Q3C_ERR_CODES = -6,              /* a code index was outside [0, codebook_size) */
    Q3C_ERR_UNSUPPORTED = -7         /* shape outside what the kernels cover */
} q3c_status;

typedef enum q3c_precision { Q3C_PRECISION_F32 = 0, Q3C_PRECISION_BF16 = 1 } q3c_precision;
typedef enum q3c_frontend { Q3C_FRONTEND_SPLIT_RVQ = 0, Q3C_FRONTEND_MEAN_EMBEDDING = 1 } q3c_frontend;

/* Mirrors the decoder_config the Python side loads (sizes of sib configuration :538-554 plus the split-RVQ
 * front end of mimi :1084-1097).  struct_size must be sizeof(q3c_config). */
typedef struct q3c_config {
    int32_t struct_size;
    int32_t frontend;                 /* q3c_frontend */
    int32_t precision;                /* q3c_precision: storage + MFMA operand type of the whole decode */
    int32_t codebook_size, num_quantizers, num_semantic_quantizers, codebook_dim, rvq_out_dim, pre_conv_kernel;
    int32_t latent_dim, hidden_size, io_proj;
    int32_t num_attention_heads, num_key_value_heads, head_dim, intermediate_size, num_hidden_layers, sliding_window;
    int32_t num_upsampling_ratios; int32_t upsampling_ratios[Q3C_MAX_STAGES];
    int32_t num_upsample_rates;    int32_t upsample_rates[Q3C_MAX_STAGES];
    int32_t decoder_dim;
    int32_t num_dilations;         int32_t dilations[Q3C_MAX_STAGES];
    int32_t max_positions;            /* longest single forward, in frames */
    float rvq_eps, rms_norm_eps, convnext_eps, snake_eps;
} q3c_config;

typedef struct q3c_ctx q3c_ctx;

/* Stages whose output q3c_set_tap can capture (as fp32, channels-last). */
typedef enum q3c_stage {
    Q3C_STAGE_RVQ_OUT = 0,          /* [B,T,rvq_out_dim]  after the output_proj of both groups (split_rvq) */
    Q3C_STAGE_TRANSFORMER_IN = 1,   /* [B,T,latent_dim]   after pre_conv / the mean embedding */
    Q3C_STAGE_TRANSFORMER_OUT = 2,  /* [B,T,latent_dim] */
    Q3C_STAGE_UPSAMPLE0 = 3,        /* + i: output of upsample stage i */
    Q3C_STAGE_CONV_IN = 16,         /* [B,4T,decoder_dim] WITH the first block's SnakeBeta already applied */
    Q3C_STAGE_BLOCK0 = 17           /* + i: output of decoder block i WITH the next SnakeBeta already applied */
} q3c_stage;

/* -- lifetime -------------------------------------------------------------------------------------------
 * Stands in for constructing the decoder module inside Qwen3TTSModel.from_pretrained (reference models.py:43-48). */
int q3c_create(const q3c_config* cfg, q3c_ctx** out);
void q3c_destroy(q3c_ctx* ctx);
const char* q3c_last_error(const q3c_ctx* ctx);
const char* q3c_version(void);

/* -- weights ---------------------------------------------------------------------------------------------
 * Stands in for load_state_dict: one call per tensor, HOST fp32 data in the PyTorch layout, names as in the
 * sibling's state dict (e.g. "decoder.1.block.2.conv1.conv.weight"); q3c_finalize() checks every tensor the
 * configuration needs is present with the right shape, repacks ([tap][out][in], transposed-conv phase split,
 * codebook = embed_sum / clamp(cluster_usage, eps) with an IEEE divide - mimi :980-983) and uploads. */
int q3c_load_weight(q3c_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
int q3c_finalize(q3c_ctx* ctx);

/* -- the hot path ----------------------------------------------------------------------------------------
 * q3c_forward = Code2Wav.forward (sib :3672-3684): codes [B][Q][T] int64 on the device ->
 * wav [B][T*total_upsample - trim] fp32 on the device, clamped to [-1, 1].  workspace is caller-owned device
 * memory of at least q3c_workspace_bytes(B, T) bytes, 256-byte aligned. */
int q3c_workspace_bytes(q3c_ctx* ctx, int32_t batch, int32_t frames, size_t* bytes);
int q3c_output_length(const q3c_ctx* ctx, int32_t frames, int64_t* samples);
int q3c_forward(q3c_ctx* ctx, const int64_t* codes_dev, int32_t batch, int32_t frames, float* wav_dev,
                void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* q3c_forward for one chunk of chunked_decode (sib :3686-3696): the chunk's first skip_frames frames are its recomputed left
 * context, whose samples the caller drops (`wav_chunk[..., context * total_upsample:]`, :3694).  Same arguments and the same
 * wav layout as q3c_forward; the samples from skip_frames * total_upsample on are bit for bit those q3c_forward writes, the
 * samples before are NOT written.  The conv decoder is causal with a finite look-back per layer, so each of its launches
 * starts at the first row a kept sample can see and the rest of the context's rows (24 of 25 frames at the sample rate) is
 * never computed; the frame-rate stages run whole.  0 <= skip_frames < frames.  While a stage tap is set the whole forward
 * runs. */
int q3c_forward_skip(q3c_ctx* ctx, const int64_t* codes_dev, int32_t batch, int32_t frames, int32_t skip_frames, float* wav_dev,
                     void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* q3c_forward_skip whose kept samples go straight to their places in a larger output (the wrapper's `wavs[i]` of reference
 * server.py:379-389 are slices of one tensor; chunked_decode's torch.cat, sib :3695-3696, becomes a placement): spans_dev is a
 * DEVICE array [batch][2] of int64 (destination offset into out_dev in samples, count); item b's samples skip_frames *
 * total_upsample .. + count - 1 of its forward are written to out_dev + offset and nothing else is written, so count <=
 * frames * total_upsample - trim - skip_frames * total_upsample, and a count of 0 drops the item (a padding row).  Same
 * arithmetic, hence the same bits, as q3c_forward_skip. */
int q3c_forward_scatter(q3c_ctx* ctx, const int64_t* codes_dev, int32_t batch, int32_t frames, int32_t skip_frames, const int64_t* spans_dev,
                        float* out_dev, void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* q3c_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  Same result as q3c_forward, bit for bit, but the workspace is
 * a static device array of Q3C_STATIC_SCRATCH_BYTES inside the library (each GPU the library runs on holds its own copy
 * until the process ends; its address on the context's device is looked up once, in q3c_finalize), and the call neither
 * allocates nor frees device memory.  Inputs and outputs are still the caller's.  A batch x frames whose
 * q3c_workspace_bytes exceeds Q3C_STATIC_SCRATCH_BYTES is refused with Q3C_ERR_WORKSPACE; per-kernel profiling must be off
 * (Q3C_ERR_STATE).  HOW MUCH FITS: the workspace grows by about 1.5 MB per frame of the batch at the production widths in bf16
 * (four arenas of frames x 1920 rows x 96 channels), so 64 MiB hold ABOUT 40 FRAMES IN ALL (3.2 s of audio: one call of
 * batch 1 x 40 frames, or 4 x 10) - a low-latency first chunk, not a bulk decode; ask q3c_workspace_bytes before relying on a shape.  Nothing in the workspace needs to start at zero - every region is written before it is read - so the
 * previous call's contents are simply overwritten; but all contexts on one GPU share the one array: calls that overlap in
 * time on the same GPU (two streams, two contexts, two threads) race and must be serialised by the caller. */
#define Q3C_STATIC_SCRATCH_BYTES (64u << 20)
int q3c_forward_static(q3c_ctx* ctx, const int64_t* codes_dev, int32_t batch, int32_t frames, float* wav_dev, void* hip_stream);

/* RVQ gather-and-sum alone (mimi :1070-1077 per group, bit-exact fp32): codes [B][Q][T] ->
 * sums [B][T][G][codebook_dim] with G = 1 (semantic) + 1 (acoustic, when num_quantizers > num_semantic). */
int q3c_rvq_dequant_f32(q3c_ctx* ctx, const int64_t* codes_dev, int32_t batch, int32_t frames, float* sums_dev,
                        void* hip_stream);

/* The ENCODE direction of the same code books (mimi :1112-1126, :1053-1068, :985-1002: the first step of the tokenizer's encode
 * side, which the reference reaches through create_voice_clone_prompt, server.py:228-232): embeddings [B][T][in_dim] fp32 on the
 * device -> codes [B][Q][T] int64.  Each group applies its bias-free 1x1 input_proj (loaded when the state dict has
 * "quantizer.<group>.input_proj.weight" [codebook_dim][in_dim][1]; otherwise in_dim must equal codebook_dim), then every layer
 * takes the nearest code-book row (smallest true squared distance, lowest index on ties) and passes the residual on; both groups
 * encode the same embeddings.  Asynchronous on the stream. */
int q3c_rvq_encode(q3c_ctx* ctx, const float* emb_dev, int32_t batch, int32_t frames, int32_t in_dim, int64_t* codes_dev,
                   void* hip_stream);

/* -- the tokenizer's ENCODE side: 24 kHz audio -> code frames -------------------------------------------------------
 * Stands in for tokenizer.encode(ref_audio), which the reference reaches through create_voice_clone_prompt (server.py:228-232,
 * 298-303) in its in-context voice-clone mode; its server keeps that mode off (x_vector_only_mode=True, server.py:221, 283)
 * and the arithmetic lives in the absent qwen-tts package, so sizes, tensor names and arithmetic follow the Mimi encoder of
 * the sibling (modeling_mimi.py: conv encoder :450-492, residual block :408-447, transformer :729-779, stride-2 conv
 * :1197-1206, _encode_frame :1230-1263).  The speaker x-vector of the reference's default mode is another network of
 * qwen-tts and is not covered.
 *
 * q3c_encoder_configure goes between q3c_create and q3c_finalize; the context then also needs the encoder's tensors
 * ("encoder.layers.N...", "encoder_transformer.layers.N...", "downsample.conv.weight", the quantiser's two input_proj, and
 * "encoder_rope.inv_freq" [head_dim / 2]).  ratios[] are the conv encoder's strides in the order it applies them. */
typedef struct q3c_encoder_config {
    int32_t struct_size;
    int32_t num_filters, kernel_size, residual_kernel_size, compress, last_kernel_size;
    int32_t num_ratios; int32_t ratios[Q3C_MAX_STAGES];
    int32_t hidden_size, num_attention_heads, head_dim, intermediate_size, num_hidden_layers, sliding_window;
    int32_t downsample_stride;        /* kernel = 2 * stride, replicate padding, no bias */
    int32_t max_positions;            /* longest audio, in rows at the transformer's rate (samples / prod(ratios)) */
    float norm_eps;
} q3c_encoder_config;
int q3c_encoder_configure(q3c_ctx* ctx, const q3c_encoder_config* cfg);
/* frames = ceil over every stride in turn (each strided conv pads its input on the right to a whole stride, mimi :269-279) */
int q3c_encode_frames(const q3c_ctx* ctx, int64_t samples, int64_t* frames);
int q3c_encode_workspace_bytes(q3c_ctx* ctx, int32_t batch, int32_t samples, size_t* bytes);
/* audio_dev [B][samples] fp32 (mono, every item the same length) -> codes_dev [B][Q][frames] int64; embeddings_dev (optional,
 * [B][frames][hidden_size] fp32) receives what the quantiser encodes.  Asynchronous on the stream; workspace as q3c_forward. */
int q3c_encode(q3c_ctx* ctx, const float* audio_dev, int32_t batch, int32_t samples, int64_t* codes_dev, float* embeddings_dev,
               void* workspace, size_t workspace_bytes, void* hip_stream);

/* Sticky device-side error word.  Synchronises the stream; if set, clears it and returns Q3C_ERR_CODES (a code index
 * outside [0, codebook_size): stands in for the IndexError torch's embedding raises) or Q3C_ERR_INVALID_ARGUMENT
 * (streaming: bad slot id, stream longer than max_positions). */
int q3c_check_errors(q3c_ctx* ctx, void* hip_stream);

/* Capture a stage output of the following q3c_forward calls as fp32 into caller memory (NULL removes the tap). */
int q3c_set_tap(q3c_ctx* ctx, int32_t stage, float* dev_buffer, size_t capacity_elements);

/* -- streaming decode ---------------------------------------------------------------------------------------
 * The reference decodes whole utterances and sends one SSE `audio` event per request (server.py:245-251, 299-335);
 * the sibling's only incremental form is chunked_decode, which recomputes 25 frames of context per chunk
 * (sib :3686-3696).  Every layer of the decode is causal, so a stream can instead be advanced n frames at a time and
 * produce exactly the samples of the un-chunked forward (sib :3672-3684) of all its frames so far: the rows each layer
 * looks back into (conv taps - the causal padding of sib :3180-3213, the previous input row of the transposed convs
 * :3216-3228, the 72-frame attention window :3267-3336, following the padding-cache pattern of mimi :73-164) are
 * carried between pushes in a per-slot state record in device memory; the attention window's keys and values are per-layer
 * ring buffers inside that record (frame f at row f mod (window - 1)), read in place and only appended to.
 *
 * A pool holds `capacity` slots.  q3c_stream_reset zeroes the given slots (a new stream starts at frame 0).
 * q3c_stream_push advances S distinct slots (ids in DEVICE memory) by n frames each in one launch sequence, whatever
 * their positions: codes [S][Q][n] int64 -> wav [S][n*total_upsample] fp32.  Over a stream's life the concatenated
 * pushes are `trim` blank samples (zeros, at the start of its first push) followed by the un-chunked forward's output
 * for the frames pushed so far.  Errors found on the device (slot id outside the pool, more than max_positions
 * frames in one stream) are reported by q3c_check_errors.  q3c_stream_workspace_bytes(S, n) sizes the workspace of such a push
 * AND prepares it (a small descriptor table in device memory, allocated once per (S, n) pair); q3c_stream_push itself never
 * allocates or synchronises and returns Q3C_ERR_STATE for a pair that has not been prepared.  Destroy pools before their
 * context.
 *
 * q3c_stream_push_graph is the same push replayed from a HIP graph (about 90 small launches become one graph launch).  A
 * graph's kernel nodes hold their operand addresses, so graphs are cached per (S, n, slots_dev, codes_dev, wav_dev,
 * workspace_dev): the first call with a new combination records the launch sequence on a side stream and instantiates it
 * (host work of a few milliseconds), later calls only launch it; callers therefore keep a fixed set of buffers and refill
 * them.  Up to 32 combinations per (S, n) are cached, further ones run as a plain push; while per-kernel profiling or a tap
 * is on, or while the caller's own stream is capturing, it is a plain push too. */
typedef struct q3c_stream_pool q3c_stream_pool;
int q3c_stream_pool_create(q3c_ctx* ctx, int32_t capacity, q3c_stream_pool** out);
void q3c_stream_pool_destroy(q3c_stream_pool* pool);
int q3c_stream_state_bytes(const q3c_stream_pool* pool, size_t* bytes_per_slot);
int q3c_stream_reset(q3c_stream_pool* pool, const int32_t* slots_dev, int32_t count, void* hip_stream);
int q3c_stream_workspace_bytes(q3c_stream_pool* pool, int32_t streams, int32_t frames, size_t* bytes);
int q3c_stream_push(q3c_stream_pool* pool, const int32_t* slots_dev, int32_t streams, const int64_t* codes_dev, int32_t frames,
                    float* wav_dev, void* workspace_dev, size_t workspace_bytes, void* hip_stream);
int q3c_stream_push_graph(q3c_stream_pool* pool, const int32_t* slots_dev, int32_t streams, const int64_t* codes_dev, int32_t frames,
                          float* wav_dev, void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* -- per-kernel timing (no counterpart in the reference; used by bench.py for the roofline figures) -----------
 * While enabled, every kernel launch of q3c_forward is bracketed by a pair of HIP events on the caller's stream.
 * q3c_profile_read synchronises those events and returns, per kernel, the number of launches, the summed duration,
 * and the ALGORITHMIC work of those launches: flops = 2*B*M*N*K*taps of the layer (no tile padding), bytes = the
 * tensors a memory-bound kernel must read and write once. */
typedef struct q3c_profile_entry {
    char kernel[64];
    int64_t launches;
    double ms, flops, bytes;
} q3c_profile_entry;
int q3c_profile_enable(q3c_ctx* ctx, int32_t on);   /* 0 off, 1 per kernel, 2 per (kernel, layer shape) */
int q3c_profile_read(q3c_ctx* ctx, q3c_profile_entry* out, int32_t capacity, int32_t* count);

/* -- measurement aids (no counterpart in the reference; used by tools/ and by parity tests that compare two kernels for one
 *    operation).  Both are state of ONE context: other contexts in the process are not affected. -------------------------
 * q3c_debug_set_tuning: switches that make the following launches of this context choose another tile shape or kernel than
 * the default (0).  Bits 0-7, 16-19: tile-shape choices of the contraction kernel (csrc/convgemm_impl.h); 64: the 384-channel
 * residual unit as one launch; 256: the vector-ALU attention kernel for every case; 512: the final conv as its own launch;
 * 1 << 20: the vector-ALU code-book search; 1 << 22: four-wave 256-row tiles for the 96-channel residual unit where the default
 *    is twelve-wave 768-row tiles (bit-identical results).  Every choice computes the same operation.
 * q3c_debug_set_timeline: every 64th workgroup of this context's contraction launches stores clock values at its phase
 * boundaries into dev_buffer (8 x int64 per slot); NULL switches it off. */
int q3c_debug_set_tuning(q3c_ctx* ctx, int32_t bits);
int q3c_debug_set_timeline(q3c_ctx* ctx, long long* dev_buffer, int32_t slots);
/* The same two settings for the TEMPORARY contexts the single-operator entry points below build per call (they take no
 * context argument); contexts made by q3c_create never read them. */
int q3c_debug_set_operator_context(int32_t tuning, long long* timeline_dev, int32_t slots);

/* -- single operators (host weights in the PyTorch layout are repacked and uploaded per call: meant for
 *    operator-level parity tests and integration bring-up, not for the serving loop; they synchronise) ------
 * x / y are DEVICE fp32, channels-last.  precision picks the MFMA operand type as in q3c_config. */
/* CausalConvNet (sib :3180-3213), groups = 1: x [B][L][Cin] -> y [B][L][Cout]; weight [Cout][Cin][k]. */
int q3c_causal_conv1d(const float* x_dev, int32_t batch, int32_t length, int32_t cin, int32_t cout, int32_t k, int32_t dilation,
                      const float* weight_host, const float* bias_host, float* y_dev, int32_t precision, void* hip_stream);
/* CausalTransConvNet (sib :3216-3228), k == stride or k == 2*stride: x [B][L][Cin] -> y [B][Lout][Cout];
 * weight [Cin][Cout][k]; Lout = (L-1)*stride - k + 2*stride. */
int q3c_causal_tconv1d(const float* x_dev, int32_t batch, int32_t length, int32_t cin, int32_t cout, int32_t k, int32_t stride,
                       const float* weight_host, const float* bias_host, float* y_dev, int32_t precision, void* hip_stream);
/* SnakeBeta (sib :3566-3580): y = x + 1/(exp(beta)+eps) * sin(x*exp(alpha))^2, x [rows][C].  precision BF16 runs the form the
 * bf16 decode uses (bf16 storage on both sides, range-reduced hardware sine), F32 the parity path's polynomial sine. */
int q3c_snake_beta(const float* x_dev, int64_t rows, int32_t channels, const float* alpha_host, const float* beta_host,
                   float eps, float* y_dev, int32_t precision, void* hip_stream);

/* DecoderResidualUnit (sib :3592-3608): y = x + conv_k1(snake2(conv_k7,dilation(snake1(x)))), x / y [B][L][C], C % 8 == 0;
 * conv weights [C][C][7] and [C][C][1].  Runs the launch the decode uses for the unit: one fused launch when the
 * channel count allows (<= 192 in bf16, <= 96 in fp32), otherwise two. */
int q3c_residual_unit(const float* x_dev, int32_t batch, int32_t length, int32_t channels, int32_t dilation,
                      const float* alpha1_host, const float* beta1_host, const float* w1_host, const float* b1_host,
                      const float* alpha2_host, const float* beta2_host, const float* w2_host, const float* b2_host,
                      float snake_eps, float* y_dev, int32_t precision, void* hip_stream);
/* ConvNeXtBlock (sib :3231-3263): depthwise causal conv k7 [C][1][7] -> LayerNorm(C, ln_eps) -> Linear(C,4C) -> exact GELU
 * -> Linear(4C,C) -> gamma -> + x; x / y [B][L][C], C % 8 == 0. */
int q3c_convnext_block(const float* x_dev, int32_t batch, int32_t length, int32_t channels, const float* dw_weight_host,
                       const float* dw_bias_host, const float* ln_weight_host, const float* ln_bias_host,
                       const float* pw1_weight_host, const float* pw1_bias_host, const float* pw2_weight_host,
                       const float* pw2_bias_host, const float* gamma_host, float ln_eps, float* y_dev, int32_t precision,
                       void* hip_stream);
/* Code2WavTransformerModel (sib :3454-3539) with the context's weights: hidden states x [B][T][latent_dim] fp32 ->
 * y of the same shape (input projection, sliding-window layers, final RMSNorm, output projection; positions start
 * at 0).  Asynchronous on the stream like q3c_forward; workspace as for q3c_forward(B, T). */
int q3c_transformer(q3c_ctx* ctx, const float* x_dev, int32_t batch, int32_t frames, float* y_dev, void* workspace_dev,
                    size_t workspace_bytes, void* hip_stream);

/* -- output format -------------------------------------------------------------------------------------------
 * The reference packs every waveform as a PCM_16 WAV file before it leaves the server (server.py:108-113, 117-121:
 * soundfile.write(buffer, wav, sample_rate, format="WAV"), whose default subtype is PCM_16).  This is that conversion on
 * the device - libsndfile's float -> short rule, lrintf(x * 32767) with round-half-to-even, saturated outside [-1, 1] -
 * so that half the bytes cross PCIe.  Asynchronous on the stream; wav and pcm are device pointers. */
int q3c_wav_to_pcm16(const float* wav_dev, int64_t samples, int16_t* pcm_dev, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* Q3C_H */
