/*
 * wseg.h — C-ABI of libwseg.so, the MI355X (gfx950) hot path of whisperseg_amd.
 *
 * The reference (nianlonggu/WhisperSeg) has no FFI of its own: its boundary is Python duck-typing
 * inside model.py.  Each entry point below therefore cites the reference call (file:line, relative to
 * the reference repo root) whose device work it replaces; the Python host side in
 * whisperseg_amd/{audio_utils,model,engine}.py binds these with ctypes and mirrors the reference's
 * class/method surface.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every pointer documented "device" is a HIP device
 *     pointer owned by the caller (PyTorch-ROCm allocates; the library never allocates or frees device
 *     memory and never takes ownership).  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - Every function returns 0 on success, a negative wseg_status otherwise; wseg_last_error()
 *     returns a thread-local message for the last failure.
 *   - Functions are thread-compatible: distinct models / distinct streams may be used concurrently
 *     from distinct host threads (the reference calls its backend from one Python thread per device,
 *     model.py:173-184).
 */
#ifndef WSEG_H
#define WSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSEG_ABI_VERSION 5

typedef enum {
  WSEG_OK = 0,
  WSEG_ERR_INVALID = -1,    /* bad argument / unsupported geometry */
  WSEG_ERR_HIP = -2,        /* a HIP runtime call or kernel launch failed */
  WSEG_ERR_STATE = -3,      /* model not fully populated, workspace too small, ... */
  WSEG_ERR_NO_DEVICE = -4   /* no gfx950 device visible */
} wseg_status;

typedef enum {
  WSEG_F32 = 0,             /* exact-parity mode: fp32 storage; GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32 are
                             * k-ordered fmaf chains bit for bit: tested identical to the VALU kernels they replaced) */
  WSEG_BF16 = 1,            /* bfloat16 storage + MFMA, fp32 accumulation */
  WSEG_F16 = 2,             /* IEEE half storage + MFMA, fp32 accumulation (reference WhisperSegmenterFast: CT2 float16, model.py:691) */
  /* Split-precision modes (the reference's own arithmetic is fp32 everywhere, model.py:655-666): fp32 storage and fp32
   * arithmetic for everything but the GEMMs; GEMM operands (activations and weights) are carried as hi + lo 16-bit pairs
   * and multiplied as hi*hi + hi*lo + lo*hi on the 16-bit matrix cores with fp32 accumulation (3 MFMAs per product,
   * ~2^-16 (bf16) / ~2^-21 (half) relative operand error instead of 2^-8 / 2^-11).  Weight matrices ("*.w", "dec.tok")
   * are attached pre-split: rows of 2K 16-bit words, every 32 logical columns as [32 hi | 32 lo]
   * (whisperseg_amd/engine.py::split_operand); every other tensor is float32. */
  WSEG_BF16X3 = 3,
  WSEG_F16X3 = 4,
  /* Mixed split-precision mode: WSEG_F16X3 everywhere outside the GEMMs; a GEMM takes hi*hi on the IEEE-half matrix cores and the
   * two cross terms hi*lo + lo*hi on the block-scaled MX matrix cores (fp6 e2m3 operands with per-32-element e8m0 scales, 4x the
   * 16-bit rate): 3.25 instead of 6 matrix-core issue units per 64 logical columns at a ~2^-15 operand error.  Weight
   * matrices are attached as "M6 rows" produced by wseg_convert_operand from the WSEG_F16X3 rows.  PARITY (r06): rows
   * identical to the reference's on the 200 recordings the mode was designed on, OUTSIDE the north-star tolerance on 10 of 6 000
   * held-out recordings — a fast mode; the Python layer defaults to WSEG_F16X3 (all 6 200 sweep recordings identical, like WSEG_F32). */
  WSEG_F16M6 = 5
} wseg_dtype;

int wseg_abi_version(void);
const char* wseg_last_error(void);
/* Name / gcnArch of the current device, or an error if none is visible. */
int wseg_device_info(char* name_out, size_t name_cap, int* cu_count_out);

/* ------------------------------------------------------------------------------------------------
 * Log-mel front-end.
 * Replaces, per window: reference audio_utils.py:45-76 (WhisperSegFeatureExtractor) ->
 * HF feature_extraction_whisper.py:105-133 (_np_extract_fbank_features) -> HF audio_utils.py:809-1017
 * (spectrogram), and the window slicing / zero padding / column truncation / min-fill of
 * reference model.py:140-161 — all windows of a recording in one launch pair.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_fft;            /* 512 | 1024 | 2048 | 4096 | 8192  (reference audio_utils.py:32-43) */
  int32_t hop;              /* int(spec_time_step * sr)          (reference audio_utils.py:48)    */
  int32_t n_mels;           /* 80 */
  int32_t n_cols;           /* total_spec_columns, 1000          (reference model.py:153)          */
  const float* window;      /* device [n_fft]      periodic Hann                                   */
  const float* twiddle;     /* device [n_fft/2][2] (cos, -sin)(2*pi*k/n_fft)                        */
  const int32_t* mel_start; /* device [n_mels]     first non-zero FFT bin of each filter           */
  const int32_t* mel_count; /* device [n_mels]     number of consecutive non-zero bins             */
  const int32_t* mel_offset;/* device [n_mels]     offset of the filter's weights in mel_weight    */
  const float* mel_weight;  /* device [sum(mel_count)] slaney triangle weights                     */
} wseg_logmel_desc;

/* Bytes of float scratch needed by wseg_logmel_f32 for n_windows windows of win_len samples. */
size_t wseg_logmel_scratch_bytes(const wseg_logmel_desc* d, int32_t n_windows, int64_t win_len);

/*
 * audio      device float32 [n_audio]   the whole recording (mono), resident in HBM
 * win_start  device int64   [n_windows] first sample of each window relative to audio[0]; may be
 *                                        negative (multi-trial left padding, reference model.py:138-143)
 *                                        or run past the end (zero-filled, reference model.py:150)
 * win_len    samples per window = int(total_spec_columns * spec_time_step * sr)  (model.py:133)
 * out        device float32 [n_windows][n_mels][n_cols]
 */
int wseg_logmel_f32(const wseg_logmel_desc* d, const float* audio, int64_t n_audio,
                    const int64_t* win_start, int32_t n_windows, int64_t win_len,
                    void* scratch, size_t scratch_bytes, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rational polyphase resampler (audio ingest).
 * Replaces the resampling half of `librosa.load(path, sr=target)` at reference scripts/segment.py:48,61 and
 * evaluate.py:58.  y[m] = sum_k taps[(m + pre_remove) * down - pre_pad - k * up] * x[k], all device pointers.
 * The Kaiser-windowed-sinc taps and the two alignment integers are produced by whisperseg_amd/resample.py.
 * ---------------------------------------------------------------------------------------------- */
int wseg_resample_f32(const float* x, int64_t n_in, const float* taps, int32_t n_taps, int32_t up, int32_t down,
                      int32_t pre_pad, int32_t pre_remove, float* y, int64_t n_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whisper encoder-decoder.
 * Replaces HF WhisperForConditionalGeneration as the reference drives it:
 *   construction  reference model.py:626-644 (WhisperSegmenter.__init__, from_pretrained)
 *   generate      reference model.py:647-676 / 604-622 (model.generate(...) per batch)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t d_model, n_heads, enc_layers, dec_layers, ffn, vocab;
  int32_t n_mels;           /* 80   */
  int32_t spec_cols;        /* 1000 */
  int32_t enc_positions;    /* 500 = spec_cols / 2 (reference model.py:79) */
  int32_t dec_positions;    /* 448  */
  int32_t dtype;            /* wseg_dtype: arithmetic/storage type of weights and activations */
} wseg_model_config;

typedef struct wseg_model wseg_model;

int wseg_model_create(const wseg_model_config* cfg, wseg_model** out);
void wseg_model_destroy(wseg_model* m);
/*
 * Attach one prepared weight tensor (device pointer, caller-owned, must outlive the model).
 * Names and layouts are listed in whisperseg_amd/engine.py::prepare_weights (e.g. "enc.0.qkv.w"
 * = [3*d_model, d_model] row-major in the model dtype).  bytes is checked against the expected size.
 */
int wseg_model_set_tensor(wseg_model* m, const char* name, const void* dev_ptr, size_t bytes);
/* 0 when every tensor the geometry needs has been attached. */
int wseg_model_ready(const wseg_model* m);

/* Workspace (device bytes) for max_windows window slots (= windows decoded concurrently; the encoder runs over at most 256 of
 * them per pass).
 *
 * The decoder's self-attention K / V are PAGED (ABI 5): pages of 8 positions are handed to a slot as its sequence grows and taken
 * back when its window retires, from a pool carved out of the workspace.  wseg_workspace_bytes sizes that pool for the EXPECTED
 * sequence length — min(max_length, 64) positions per slot on average — so the reference's default max_length = 448
 * (model.py:406-409) no longer reserves 448 positions per slot up front (1.3 MiB per position per slot in the split-precision
 * modes); wseg_workspace_bytes_kv takes the average number of positions per slot to provision (0: that default; >= max_length:
 * every slot can reach max_length at once, the pre-ABI-5 behaviour).  wseg_generate uses whatever the caller's workspace has room
 * for (at most a full set, at least one slot's worth); when the pool runs short it preempts the youngest window and decodes it
 * again later (same tokens; wseg_generate_stats.n_preemptions). */
size_t wseg_workspace_bytes(const wseg_model* m, int32_t max_windows, int32_t num_beams, int32_t max_length);
size_t wseg_workspace_bytes_kv(const wseg_model* m, int32_t max_windows, int32_t num_beams, int32_t max_length,
                               int32_t kv_positions_per_slot);

/*
 * Encoder only: feats device float32 [n_windows][80][1000] -> enc_out device [n_windows][500][d_model]
 * in the model dtype.  (HF modeling_whisper.py:592-646 as reached from reference model.py:655.)
 */
int wseg_encode(wseg_model* m, const float* feats, int32_t n_windows,
                void* workspace, size_t workspace_bytes, void* enc_out, void* stream);

typedef struct {
  int32_t prompt[8];              /* decoder_input_ids, reference model.py:656 */
  int32_t prompt_len;
  int32_t eos_token_id, pad_token_id;   /* reference model.py:658-659 */
  int32_t max_length;             /* total length incl. prompt (HF 4.38.2 semantics), model.py:660 */
  int32_t num_beams;              /* 1 = greedy (do_sample with top_k=1, model.py:662-663), else beam search */
  float length_penalty;           /* model.py:665 */
  const int32_t* suppress_tokens; /* device [n_suppress]   generation_config.suppress_tokens        */
  int32_t n_suppress;
  const int32_t* begin_suppress_tokens; /* device [n_begin_suppress] applied at the first generated position */
  int32_t n_begin_suppress;
  /* In-flight batching (the reference decodes batch by batch, model.py:653, one file at a time,
   * scripts/segment.py:39-56).  0 selects the default of each. */
  int32_t n_slots;                /* window slots decoded concurrently; 0 or >= n_windows: one per window                */
  int32_t refill_min;             /* admit queued windows once this many slots are free; 0: n_slots / 8 (min 1)         */
  int32_t lookahead;              /* decode steps the host may run ahead of the device; 0: 1                            */
  const int32_t* window_max_length; /* device [n_windows] per-window cap on the total length (clamped to max_length), or
                                     NULL: every window may run to max_length                                           */
  /* Sampling (reference model.py:615-616, 662-663: do_sample = (num_beams == 1) with top_k / top_p).  Used only when
   * num_beams == 1 and top_k > 1: the next token is drawn from softmax over the top_k processed logits, further cut to the
   * nucleus top_p (HF TopKLogitsWarper then TopPLogitsWarper), with a counter-based generator keyed by (seed, window,
   * position) — reproducible for a seed, but not the torch generator stream HF would consume.  top_k <= 1: argmax. */
  int32_t top_k;                  /* 0 / 1: deterministic argmax (the reference's default top_k = 1); 2..16: sample       */
  float top_p;                    /* nucleus mass in (0, 1]; values <= 0 or >= 1 disable the cut                          */
  uint64_t seed;
  const void* encoder_output;     /* device [n_windows][enc_positions][d_model] in the model dtype: precomputed encoder
                                     states (wseg_encode) used instead of running the encoder on feats (feats may then
                                     be NULL; rows past the last window must be readable up to a multiple of 256), or NULL */
} wseg_generate_params;

/*
 * Full decode of n_windows windows: encoder, cross-K/V, then greedy / beam search with HF semantics
 * (generation/utils.py:3208-3510), entirely on device.
 *   out_tokens  device int32 [n_windows][max_length]  best sequence INCLUDING the prompt, pad-filled
 *   out_lengths device int32 [n_windows]              number of valid tokens (prompt + generated)
 * The windows are decoded through n_slots window slots: a slot whose window has finished (EOS / early-stop heuristic /
 * max_length) is retired and re-used for the next queued window while the other slots keep decoding, every slot at its
 * own position.  Idle slots are skipped by every per-step kernel.  The workspace is sized by
 * wseg_workspace_bytes(m, n_slots, ...): it does not grow with n_windows.  The host stays `lookahead` steps ahead of the
 * device and otherwise only waits on the small per-step status mirror; the call is stream-ordered on `stream`.
 * (ABI 3 had decode "lanes" — slot groups on separate streams; they measured exactly as one group with their total slot
 * count and were removed in ABI 4.  ABI 5: paged self-attention K / V, see wseg_workspace_bytes.)
 */
int wseg_generate(wseg_model* m, const float* feats, int32_t n_windows, const wseg_generate_params* p,
                  void* workspace, size_t workspace_bytes,
                  int32_t* out_tokens, int32_t* out_lengths, void* stream);

/* Scheduler statistics of the last wseg_generate call on this model. */
typedef struct {
  int32_t n_windows, n_slots;
  int32_t n_steps;                /* decode steps launched (each steps every active slot once)                          */
  int32_t n_admissions;           /* encoder + cross-K/V passes (groups of windows admitted into free slots)            */
  int64_t slot_steps_active;      /* sum over steps of slots that were decoding a window                                */
  int64_t slot_steps_total;       /* steps * slots                                                                      */
  int64_t queued_slot_steps_active; /* the same two sums over the steps launched while windows were still queued, i.e.   */
  int64_t queued_slot_steps_total;  /* without the drain of the last windows (steady-state occupancy of the refill)      */
  int32_t kv_units_total;         /* self-attention K/V pool of the call: units of (one 8-position page x all beams of a slot) */
  int32_t kv_units_peak;          /* most units in use at once                                                           */
  int32_t n_preemptions;          /* windows aborted for lack of pool units and decoded again later                       */
  int32_t reserved_;
} wseg_generate_stats;
int wseg_last_stats(const wseg_model* m, wseg_generate_stats* out);

/* WSEG_F16M6: GEMM operand rows of the mixed split-precision mode from WSEG_F16X3 operand rows (whisperseg_amd/engine.py::
 * split_operand with IEEE-half pairs).  src: device, rows of 2 K 16-bit words; dst: device, rows of 4 K bytes (must not alias
 * src); K % 64 == 0.  weight_order != 0 for the W operand of a GEMM (weight matrices: what wseg_model_set_tensor expects in
 * WSEG_F16M6 mode), 0 for an activation operand (only the library's own GEMM wrappers and wseg_debug_gemm need that). */
int wseg_convert_operand(const void* src_x3_rows, void* dst_m6_rows, int64_t n_rows, int32_t K, int32_t weight_order, void* stream);

/* Debug/parity taps (used by tests): first-step logits fp32 [n_windows*num_beams][vocab] of the last
 * wseg_generate call are kept in the workspace; this copies them out (device to device). */
int wseg_debug_first_logits(wseg_model* m, void* workspace, float* out, int32_t n_rows, void* stream);

/* Per-stage device time (ms) of the last wseg_generate call on this model, measured with HIP events
 * on the call's stream: [0]=encoder passes, [1]=cross-K/V passes, [2]=everything else (the decode steps),
 * [3]=number of decode steps. */
int wseg_last_timing(const wseg_model* m, float out[4]);

/* Test / tuning tap: out[M][N] = epilogue(A[M][K] * W[N][K]^T + bias) with the library's GEMM of the given dtype.
 * epi: 0 store, 1 GELU, 2 residual add (resid[M][N] and out are the fp32 residual stream in every dtype).  A must have
 * round_up(M,256) readable rows, N % 128 == 0,
 * K % 64 == 0.  Used by tests/test_gemm_gpu.py (parity vs torch / fp64) and tools/gemm_bench.py. */
int wseg_debug_gemm(int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const void* A, const void* W,
                    const void* bias, const void* resid, void* out, void* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* WSEG_F16M6: 1 when wseg_debug_gemm with epi 0 / 1 writes its output as M6 rows (the LDS-staged epilogues of the large-tile
 * kernels; the skinny family when it splits K — assumed here: a workspace that never limits the split), 0 when it writes hi | lo
 * IEEE-half rows (skinny family, K not split); always 0 for the other dtypes. */
int wseg_debug_gemm_out_is_mx(int32_t dtype, int32_t M, int32_t N, int32_t K);

/* Test / tuning tap of the decoder's fused step x += A W^T + bias; y = LayerNorm(x) * gamma + beta (x: fp32 residual stream [M][N],
 * in place; y: GEMM operand rows of the dtype; bias / gamma / beta: parameter type of the dtype).  Same operand rules as
 * wseg_debug_gemm.  Used by tests/test_gemm_gpu.py and tools/gemm_bench.py --resid-ln. */
int wseg_debug_gemm_resid_ln(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, const void* W, const void* bias, void* x,
                             const void* gamma, const void* beta, void* y, void* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* Test tap of the cross-lane exchanges every wave reduction of the library is built on (csrc/wseg_common.h lane_xor<M>: DPP
 * quad_perm / row_shl / row_shr / row_ror, v_permlane16_swap, v_permlane32_swap instead of ds_bpermute): out[m][l] = the value
 * lane l ^ (1 << m) holds, for m = 0..5 and l = 0..63, where lane l holds 7 l + 3.  out: device, 6 * 64 uint32.
 * tests/test_gemm_gpu.py::test_lane_exchanges. */
int wseg_debug_lane_xor(uint32_t* out, void* stream);

/* Live per-launch timing of the dominant kernel (the 256x256 ping-pong bf16 MFMA GEMM; 128x128 persistent for narrow problems) with HIP events recorded on
 * the launching stream.  Between begin and end every launch of that kernel is bracketed by two events;
 * end synchronises and returns the sums: algorithmic FLOPs (2*M*N*K of the real, un-padded problem),
 * kernel milliseconds and the number of launches.  Process-wide; intended for bench.py's roofline leg. */
int wseg_profile_begin(void);
int wseg_profile_end(double* total_flops, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* WSEG_H */
