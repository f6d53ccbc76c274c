// Host-side launchers of the gfx950 kernels (internal to libwseg).
#pragma once
#include "wseg_common.h"

namespace wseg {

// Decoder self-attention K / V live in PAGES of KV_PAGE positions, allocated on demand by wseg_generate's scheduler from a pool
// sized for the EXPECTED, not the maximal, sequence length (the API default max_length = 448, reference model.py:406-409, would
// otherwise cost 0.6-1.3 MiB per position per slot up front: 376 MiB per slot, against 10-60 tokens a real window emits).
// A pool UNIT = one page of every beam row of a slot, in every layer, K and V:  pool[layer][unit][beam][head][KV_PAGE][64].
// Page table: kv_pt[slot][position / KV_PAGE] -> unit (the beams of a slot advance together, so one entry serves them all).
constexpr int KV_PAGE = 8;

// ---------------------------------------------------------------------------------------------
// GEMM:  C[m][n] = sum_k A[m][k] * W[n][k]   (A [M][lda], W [N][ldw]; both K-contiguous, i.e. W is a
// torch.nn.Linear weight as stored).  Epilogues fuse bias / GELU / residual / layout scatter.
// ---------------------------------------------------------------------------------------------
enum EpiKind {
  EPI_STORE = 0,     // out[m][n] = acc + bias
  EPI_GELU,          // out = gelu(acc + bias)
  EPI_RESID,         // out = resid + acc + bias
  EPI_GELU_POS,      // out = gelu(acc + bias) + pos[m % pos_rows][n]          (conv2 + positional emb.)
  EPI_QKV_ENC,       // fused q|k|v: q*scale -> Q[b][h][t][64], k -> K[b][h][t][64], v -> Vt[b][h][64][Tp]
  EPI_KV_CROSS,      // fused k|v of one decoder layer: K[b][h][t][64], V[b][h][t][64]
  EPI_F32,           // out_f32[m][n] = acc (+ bias)
  EPI_QKV_DEC,       // decoder step: q*scale -> q[m][n]; k,v -> paged self cache [unit][beam][h][pos % KV_PAGE][64]
  EPI_SCALE,         // out = (acc + bias) * scale                                (cross-attention q)
  EPI_COUNT
};

struct EpiParams {
  const void* bias = nullptr;  // model dtype [N] (nullptr = none)
  void* out = nullptr;         // model dtype; fp32 for the residual-stream epilogues (EPI_RESID, EPI_GELU_POS)
  int ldc = 0;
  const void* resid = nullptr; // fp32 residual stream [M][ldc]
  const void* pos = nullptr;
  int pos_rows = 1;
  float scale = 1.f;
  void* q = nullptr;
  void* k = nullptr;
  void* v = nullptr;
  int d_model = 0;
  int t_len = 1;               // rows per window (500)
  int t_pad = 1;               // padded rows per (b,h) slab (512)
  int n_heads = 1;
  const int* pos_ptr = nullptr;  // device [rows / pos_div]: current decode position of each window slot (EPI_QKV_DEC)
  int pos_div = 1;               // rows (beams) per slot
  const int* kv_pt = nullptr;    // device [slots][kv_npg]: pool unit of each KV_PAGE positions of a slot (EPI_QKV_DEC)
  int kv_npg = 0;
  const int* idle_ptr = nullptr; // device [slots]: 1 = the slot is idle (EPI_QKV_DEC must not store its K / V: the page its stale
                                 // table entry names may already belong to another slot)
  const int* slot_map = nullptr; // device [M / t_len]: destination window slot of each window of the batch (EPI_KV_CROSS); null = identity
  float* out_f32 = nullptr;
  int qkv_mode = 0;              // split-precision modes, EPI_QKV_ENC storage of Q / K / V^T (x3_enc_attention_mode): 0 = IEEE half,
                                 // 1 = fp32 (fp32 attention kernel), 2 = half hi + lo planes, the lo plane qkv_plane elements behind
  size_t qkv_plane = 0;
  int vt_tiled = 0;              // EPI_QKV_ENC: V^T in the MFMA operand order of the 16-bit attention kernel (vt_tiled_index, wseg_common.h)
                                 // instead of plain [b][h][64][t_pad] rows (the fp32 attention kernels): enc_attention_vt_tiled(dtype)
  int kv24 = 0;                  // split-precision modes, EPI_KV_CROSS storage of the cross K / V (x3_cross_kv_format): 0 = fp32;
                                 // 1 = 24-bit FLOATS in two planes per (slot, head): [t_len][64] top halves (16 bits) then [t_len][64]
                                 // third bytes (knob builds; bf16x3 / f16x3 until mid r06); 2 (f16m6, r05) = block floating point, one block
                                 // per (position, head) row: [t_len][64] int16 then [t_len] fp32 powers of two, value = int16 * scale (132
                                 // bytes per row); 3 (bf16x3 / f16x3, r06) = block floating point with 24-bit integers in the two planes
                                 // of format 1, then [t_len] fp32 powers of two (196 bytes per row)
};

struct GemmArgs {
  const void* A; int lda;
  const void* W; int ldw;
  int M, N, K;
  EpiParams ep;
  float* splitk_ws = nullptr;     // fp32 [splits][M_pad][N] when the launcher decides to split K
  size_t splitk_ws_bytes = 0;
  int plan_m = 0;                 // > 0: choose the PLAN (kernel family, tile size, split-K ranges: everything that fixes the summation order of an
                                  // output element) as for plan_m rows and launch it over the M rows given.  The admission pass of the slot scheduler
                                  // runs whatever number of rows was admitted on the plan of the call's decode step, so that a window's numbers do not
                                  // depend on how many neighbours were admitted with it.  plan_m >= M.
};

// dtype: WSEG_F32 (exact kernels), WSEG_BF16 / WSEG_F16 (MFMA) or WSEG_BF16X3 / WSEG_F16X3 (split-precision MFMA: A and W
// are hi | lo operand rows, K / lda / ldw stay LOGICAL, bias / q / k / v are fp32, EPI_STORE / EPI_GELU write operand rows).
// M may be any value as long as A has round_up(M,256) readable rows; N % 128 == 0 rows of W readable; K % 64 == 0
// (K % 32 == 0 in the split-precision modes).
int launch_gemm(int dtype, EpiKind epi, const GemmArgs& g, hipStream_t s);
// the exact-parity fp32 kernels (wseg_gemm_f32.hip; reached through launch_gemm with WSEG_F32)
int launch_gemm_f32(EpiKind epi, const GemmArgs& g, hipStream_t s);
// CU count of the CURRENT device, cached per device
int device_cu_count();
// WSEG_F16M6: does an EPI_STORE / EPI_GELU launch of this (logical) shape write M6 rows (true) or hi | lo rows (false)?
// splitk_ws_bytes: the split-K workspace the launch will be given (0: none) — the skinny family writes M6 rows when it splits K
// (M: the rows the PLAN is chosen for — GemmArgs::plan_m when the launch sets it)
bool gemm_out_is_mx(int dtype, int M, int N, int K, size_t splitk_ws_bytes = 0);
// Split-K partial sums only (bf16 decoder rows): part[z][m_pad][N] fp32 in g.splitk_ws, no epilogue.  The consumer
// kernel (decoder self-/cross-attention) finishes the reduction itself.  Returns false in *ok when the shape is not
// served by the skinny family (caller falls back to launch_gemm).
struct PartialInfo { const float* part = nullptr; int splits = 0; int m_pad = 0; int n = 0; };
int launch_gemm_partial(int dtype, const GemmArgs& g, PartialInfo* info, bool* ok, hipStream_t s);
// g.ep: bias, resid == out == x, ldc == N.  x += A W^T + bias;  y = LayerNorm(x)  (fused for bf16 decoder rows)
int launch_gemm_resid_ln(int dtype, const GemmArgs& g, const void* gamma, const void* beta, void* y, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Encoder-side kernels (wseg_enc.hip)
// ---------------------------------------------------------------------------------------------
// feats f32 [B][80][1000] -> A1 [B*1000][Kp] with k = tap*80 + c (zero padded to Kp).
int launch_im2col_conv1(int dtype, const float* feats, void* a1, int B, int n_mels, int cols, int kp, hipStream_t s);
// h1 [B*1000][d] -> A2 [B*500][3d] with k = tap*d + c, stride 2, pad 1.
int launch_im2col_conv2(int dtype, const void* h1, void* a2, int B, int cols, int d, hipStream_t s);
// y[m][:] = LayerNorm(x[m][:]) * g + b, eps 1e-5; x is the fp32 residual stream, y / g / b have the model dtype.
int launch_layernorm(int dtype, const float* x, const void* g, const void* b, void* y, int M, int d, hipStream_t s);
// Encoder self-attention over Q,K [B][H][Tp][64], Vt [B][H][64][Tp] (q pre-scaled) -> out [B*T][d].
// WSEG_F16M6: *out_is_mx (may be null) reports whether out was written as M6 rows (split-precision attention) or as hi | lo rows.
bool enc_attention_writes_mx(int dtype);
// does launch_enc_attention(dtype, ...) read V^T in MFMA operand order (the 16-bit MFMA kernel) or as plain rows (the fp32 kernels)?
bool enc_attention_vt_tiled(int dtype);
int launch_enc_attention(int dtype, const void* q, const void* k, const void* vt, void* out,
                         int B, int H, int T, int Tp, int d, hipStream_t s);

// split-precision modes: arithmetic of the encoder self-attention -> EpiParams::qkv_mode.  2 (default): split precision (half
// hi + lo operands, three MFMAs per product); 0: plain IEEE half (WSEG_X3_ENC_ATTN=f16); 1: fp32 matrix cores (=f32).
int x3_enc_attention_mode();
// Storage format of the cross-attention K / V in the split-precision modes (EpiParams::kv24) and its bytes per (position, head) row.
// 1 = fp32 words rounded to their top 24 bits (sign, exponent, 15 + 1 mantissa bits — the ">= 16 bits" the precision study asks of the
// cross K; 3 instead of 4 bytes per element of an HBM-bound stream): bf16x3 / f16x3 until mid r06, now a knob-build format.
// 2 = per-row block floating point (r05): the 200-recording sweep through the CPU oracle with K and V so quantised is 200 / 200 and the
// first-step logit error stays at the mixed mode's own 1.5e-4 (24-bit: 1.5e-4; plain half: 7.5e-4, 196 / 200; tools/precision_study.py
// "ckv=bfp16r", profiles/r05_precision_study.json) for 132 instead of 192 bytes per row of an HBM-bound stream: f16m6 (wseg_dec.hip,
// x3_cross_kv_format, says why the three-MFMA modes do not take it).
int x3_cross_kv_format(int dtype, int nb);      // 0 = fp32 (more than 4 beams), 1 = 24-bit (bf16x3 / f16x3), 2 = bfp16 rows (f16m6)
// 3 (r06: bf16x3 / f16x3) = block floating point with 24-bit integers in the two-plane layout of format 1 + [t_len] fp32 row scales:
// the bytes of the 24-bit floats (+ 4 per row), ~100x their precision relative to the row maximum (st_bfp24_row, wseg_gemm_epi.h).
static inline size_t cross_kv_row_bytes(int fmt, size_t es) { return fmt == 3 ? 196 : (fmt == 2 ? 132 : (fmt == 1 ? 192 : 64 * es)); }
// WSEG_F16M6: hi | lo IEEE-half operand rows [M][2K words] -> M6 rows [M][4K bytes] (wseg_common.h), K % 64 == 0
int launch_x3_to_m6(const void* x3_rows, void* m6_rows, size_t M, int K, bool weight_order, hipStream_t s);
// the dtype every NON-GEMM kernel runs in: WSEG_F16M6 is WSEG_F16X3 outside the GEMMs
static inline int storage_dtype(int dtype) { return dtype == WSEG_F16M6 ? WSEG_F16X3 : dtype; }
// Split-precision modes only: operand rows (hi | lo pairs, wseg_common.h) [M][2d words] <-> fp32 [M][d], d % 32 == 0.
int launch_operand_to_f32(int dtype, const void* op, float* out, size_t M, int d, hipStream_t s);
int launch_f32_to_operand(int dtype, const float* in, void* op, size_t M, int d, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Decoder-side kernels (wseg_dec.hip)
// ---------------------------------------------------------------------------------------------
struct DecodeState;   // device-resident bookkeeping, defined in wseg_dec.h

}  // namespace wseg
