// Model object, workspace planning and the encode / generate drivers of libwseg.
//
// Replaces HF WhisperForConditionalGeneration as the reference drives it (reference model.py:626-676):
// encoder (HF modeling_whisper.py:592-646), decoder with KV cache (:690-796), tied LM head (:1080) and
// greedy / beam-search decoding (HF generation/utils.py:3208-3510).  The host code below only enqueues
// kernels on the caller's stream; all decoding state lives in the caller-provided workspace.
#include <algorithm>
#include <deque>
#include <map>
#include <string>
#include <vector>
#include "wseg_dec.h"

using namespace wseg;

namespace {

struct Slot { const void** field; size_t bytes; bool set; };

struct EncLayer { const void *ln1_g, *ln1_b, *qkv_w, *qkv_b, *o_w, *o_b, *ln2_g, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b; };
struct DecLayer {
  const void *ln1_g, *ln1_b, *qkv_w, *qkv_b, *o_w, *o_b, *ln2_g, *ln2_b, *cq_w, *cq_b, *ckv_w, *ckv_b, *co_w, *co_b,
      *ln3_g, *ln3_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};

struct DecPlan {   // decoder-side buffers (W window slots)
  int W = 0;
  int row_cap = 0;                           // rows the activation buffers hold (W * beams rounded up to 256)
  char *ck, *cv, *sk, *sv, *dx, *dy, *dq, *dattn, *dh, *logits, *first_logits, *splitk, *mask;
  int kv_units = 0;                          // pool units of the paged self-attention K / V (wseg_kernels.h)
  size_t kv_layer_stride = 0;                // bytes between the pools of consecutive layers (K and V alike)
  int* kv_pt = nullptr;                      // page table [W][npg]
  int* kv_pairs = nullptr;                   // scratch of the page-table update kernel [2 * W]
  char *tk_val, *tk_idx, *tk_stat;
  int *adm_slots, *adm_wins, *ret_slots;     // device lists written by the scheduler (admission / retirement)
  int* zeros = nullptr;                      // [W] zeros: the "idle" flags of the prompt pass's view of the admitted windows
  int* seed_dev;                             // sampling seed (2 words), rewritten per call: the step graph stays valid
  size_t splitk_bytes;
  DecodeState st;
};

struct Plan {   // workspace carve-up (all offsets 256-byte aligned)
  size_t total = 0;
  char *a1, *h1, *a2, *x, *y, *q, *k, *vt, *hbuf, *enc_out;
  char *mxa = nullptr, *mxe = nullptr;      // WSEG_F16M6: M6 rows of the current GEMM's activation operand / of the encoder output
  DecPlan dec;
};

// Pinned host staging for the scheduler's small host->device lists and the device->host status mirror.  An entry is
// re-used only after the event recorded behind its copy has completed.
struct PinnedRing {
  static constexpr int N = 8;
  int* host = nullptr;        // [N][cap]
  int cap = 0;
  hipEvent_t ev[N] = {};
  bool used[N] = {};
  int next = 0;
};

// Scheduler state of wseg_generate that survives between calls: pinned staging rings, timing events, the captured
// decode-step graph.
struct Sched {
  std::vector<hipEvent_t> ev_pool;     // timing events: pairs around every encoder / cross-K/V pass
  size_t ev_used = 0;
  std::vector<int> ev_enc, ev_ckv;     // indices of (begin, end) pairs in ev_pool
  PinnedRing ring_h2d, ring_status;
  // decode-step graph (hipGraph): captured once per (workspace, geometry, parameters), replayed per step
  hipGraphExec_t step_graph = nullptr;
  hipStream_t cap_stream = nullptr;    // capture happens on a private stream (the legacy NULL stream cannot capture)
  std::vector<unsigned char> step_graph_key;
};

}  // namespace

struct wseg_model {
  wseg_model_config cfg;
  size_t es;                 // element size of the model dtype
  bool x3 = false;           // split-precision mode (GEMM operands are hi | lo rows, everything else fp32)
  bool mx = false;           // WSEG_F16M6: GEMM operands are M6 rows (weights attached converted; activations converted before each GEMM)
  int sdt = 0;               // dtype of every non-GEMM kernel (WSEG_F16M6 -> WSEG_F16X3)
  const void* dec_tok_f32 = nullptr;      // WSEG_F16M6: fp32 copy of the token embedding for the embedding lookup
  size_t ckv_es = 0;         // bytes per cross-attention K / V element: es, or 3 (24-bit planes) in the split modes with <= 4 beams
  int kp1, vp, tp;           // conv1 K padded, vocab padded, encoder positions padded
  std::map<std::string, Slot> slots;
  const void *conv1_w, *conv1_b, *conv2_w, *conv2_b, *enc_pos, *enc_ln_g, *enc_ln_b;
  const void *dec_tok, *dec_pos, *dec_ln_g, *dec_ln_b;
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  Sched sched;
  // last wseg_generate call: whole-call event pair, geometry, scheduler statistics
  int ev_total[2] = {-1, -1};
  bool timing_valid = false;
  int last_W = 0, last_nb = 0, last_L = 0, last_units = 0;
  bool first_logits_valid = false;
  wseg_generate_stats stats = {};
};

namespace {

void add_slot(wseg_model* m, const std::string& name, const void** field, size_t elems) {
  *field = nullptr;
  m->slots[name] = Slot{field, elems * m->es, false};
}

// The encoder (and the cross-K/V GEMMs behind it) runs over at most ENC_CHUNK windows at a time: its activations (FFN hidden:
// 10 MB per window in the 16-bit modes, 20 MB in the split modes) then stop growing with the slot count, and a pass of 256
// windows (128 000 rows: the r01 / r02 headline workload) already fills the chip for tens of rounds.
static const int ENC_CHUNK = std::max(1, WSEG_KNOB_INT("WSEG_ENC_CHUNK", 256));      // (measurement knob, variant builds)
// Default self-K/V pool of wseg_workspace_bytes: positions per slot, or max_length if that is smaller.
constexpr int KV_DEFAULT_POSITIONS = 64;

int kv_pages(int L) { return (L + KV_PAGE - 1) / KV_PAGE; }
// pool units for W slots holding `per` positions each on average — never less than ONE slot's worth of max_length (the scheduler's
// progress guarantee: a lone window can always finish)
int kv_units_for(int W, int L, int per) { return std::max(W * kv_pages(per < L ? per : L), kv_pages(L)); }
int kv_default_units(int W, int L) { return kv_units_for(W, L, KV_DEFAULT_POSITIONS); }
size_t kv_unit_bytes(const wseg_model* m, int nb) {      // one unit: every layer, K and V
  return (size_t)m->cfg.dec_layers * 2 * nb * m->cfg.n_heads * KV_PAGE * 64 * m->es;
}

// Lay out the workspace for W window slots (encoder passes of up to min(W, ENC_CHUNK) windows), nb beams, capacity L positions,
// kv_units pool units of self-attention K / V.  base may be null (size query).
void make_plan(const wseg_model* m, int W, int nb, int L, int kv_units, char* base, Plan& p) {
  const wseg_model_config& c = m->cfg;
  const size_t es = m->es;
  const size_t d = c.d_model, H = c.n_heads, ffn = c.ffn;
  const int We = W < ENC_CHUNK ? W : ENC_CHUNK;
  const size_t M1p = align_up((size_t)We * c.spec_cols, 256), Mp = align_up((size_t)We * c.enc_positions, 256);
  char* cur = base;
  auto take = [&](size_t bytes) { char* q = cur; cur += align_up(bytes, 256); return q; };
  // encoder: a1 | h1 | a2 are dead once conv2 has run; hbuf reuses their space.
  const size_t conv_bytes = align_up(M1p * m->kp1 * es, 256) + align_up(M1p * d * es, 256) + align_up(Mp * 3 * d * es, 256);
  const size_t hbuf_bytes = align_up(Mp * ffn * es, 256);
  char* u = take(conv_bytes > hbuf_bytes ? conv_bytes : hbuf_bytes);
  p.a1 = u;
  p.h1 = p.a1 + align_up(M1p * m->kp1 * es, 256);
  p.a2 = p.h1 + align_up(M1p * d * es, 256);
  p.hbuf = u;
  p.x = take(Mp * d * 4);                          // the residual stream is fp32 in every mode
  p.y = take(Mp * d * es);
  p.q = take((size_t)We * H * m->tp * 64 * es);
  p.k = take((size_t)We * H * m->tp * 64 * es);
  p.vt = take((size_t)We * H * m->tp * 64 * es);
  p.enc_out = take(Mp * d * es);
  const size_t Ld = c.dec_layers, Tk = c.enc_positions;
  if (m->mx) {
    // scratch for the activation operands that still arrive as hi | lo rows and are converted in front of their GEMM: the conv1
    // im2col always; conv2's operand, the attention output and the FFN hidden only when their producer is not one that writes
    // M6 rows directly (4-column epilogues of the skinny GEMM family on small problems; WSEG_X3_ENC_ATTN != split)
    const int dtp = c.dtype, R0 = W * nb;
    const size_t Rp0 = align_up((size_t)R0, 256);
    size_t big = M1p * (size_t)m->kp1;
    if (!enc_attention_writes_mx(dtp)) big = std::max(big, Mp * d);
    for (int n = 1; n <= We; ++n) {      // an encoder pass runs over 1 .. We windows (refills admit a few at a time)
      const size_t mp = align_up((size_t)n * c.enc_positions, 256);
      if (!gemm_out_is_mx(dtp, n * c.spec_cols, (int)d, m->kp1)) big = std::max(big, mp * 3 * d);
      if (!gemm_out_is_mx(dtp, n * c.enc_positions, (int)ffn, (int)d)) big = std::max(big, mp * ffn);
    }
    big = std::max(big, Rp0 * ffn);      // decode step / prompt pass of any row count up to R0 whose FFN hidden arrives as hi | lo rows
    if (!dec_cross_attn_writes_mx(dtp, nb)) big = std::max(big, Rp0 * d);      // 5..8 beams: cross-attention output as hi | lo rows
    p.mxa = take(big * 4);
    p.mxe = take(Mp * d * 4);
  }
  const size_t maxn = 3 * d > ffn ? 3 * d : ffn;
  DecPlan& q = p.dec;
  q.W = W;
  const size_t Wc = W, R = Wc * nb, Rp = align_up(R, 256);
  const size_t crow = cross_kv_row_bytes(m->x3 ? x3_cross_kv_format(c.dtype, nb) : 0, es);      // bytes per (position, head) row (wseg_dec.hip)
  q.ck = take(Ld * Wc * H * Tk * crow);
  q.cv = take(Ld * Wc * H * Tk * crow);
  q.kv_units = kv_units;
  q.kv_layer_stride = align_up((size_t)kv_units * nb * H * KV_PAGE * 64 * es, 256);
  q.sk = take(Ld * q.kv_layer_stride);
  q.sv = take(Ld * q.kv_layer_stride);
  q.kv_pt = (int*)take(Wc * kv_pages(L) * 4);
  q.kv_pairs = (int*)take(2 * Wc * 4);
  q.row_cap = (int)Rp;
  q.dx = take(Rp * d * 4);                         // decoder residual stream, fp32
  q.dy = take(Rp * d * es);
  q.dq = take(Rp * d * es);
  q.dattn = take(Rp * d * es);
  q.dh = take(Rp * ffn * es);
  q.logits = take(Rp * (size_t)m->vp * 4);
  q.first_logits = take(R * (size_t)m->vp * 4);
  q.splitk_bytes = (size_t)8 * Rp * maxn * 4;   // fp32 partials of the decoder-step GEMMs
  // ... and of the cross-K/V GEMM of a FEW windows (the stream kernels below the large-tile threshold of ~340 tiles of 128x128: at most
  // 128 * 340 / (N / 128) rows of N fp32 columns = 22.3 MB whatever N is): its block-floating-point row writer lives in the reduction kernel
  if (m->x3) q.splitk_bytes = std::max(q.splitk_bytes, (size_t)24 << 20);
  q.splitk = take(q.splitk_bytes);
  q.mask = take(align_up((size_t)c.vocab, 4));
  q.tk_val = take(R * 256 * 4);
  q.tk_idx = take(R * 256 * 4);
  q.tk_stat = take(R * 16 * 2 * 4);
  q.adm_slots = (int*)take(Wc * 4);
  q.adm_wins = (int*)take(Wc * 4);
  q.ret_slots = (int*)take(Wc * 4);
  q.zeros = (int*)take(Wc * 4);
  q.seed_dev = (int*)take(8);
  DecodeState& st = q.st;
  st.W = (int)Wc; st.nb = nb; st.L = L; st.V = c.vocab; st.ldv = m->vp;
  st.kv_pt = q.kv_pt; st.npg = kv_pages(L);
  st.pos = (int*)take(Wc * 4);
  st.done = (int*)take(Wc * 4);
  st.win = (int*)take(Wc * 4);
  st.wmax = (int*)take(Wc * 4);
  st.win_max_length = nullptr;
  st.tokens_in = (int*)take(R * 4);
  st.run_seq = (int*)take(R * L * 4);
  st.fin_seq = (int*)take(R * L * 4);
  st.run_score = (float*)take(R * 4);
  st.fin_score = (float*)take(R * 4);
  st.fin_flag = (int*)take(R * 4);
  st.fin_len = (int*)take(R * 4);
  st.unsat = (int*)take(Wc * 4);
  st.anc = (unsigned char*)take(R * L);
  st.cand_val = (float*)take(R * MAX_CAND * 4);
  st.cand_tok = (int*)take(R * MAX_CAND * 4);
  st.sup_mask = (const unsigned char*)q.mask;
  p.total = (size_t)(cur - base);
}

int check_geometry(const wseg_model_config& c) {
  if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model != c.n_heads * 64) { set_error("d_model %d must be n_heads %d * 64", c.d_model, c.n_heads); return WSEG_ERR_INVALID; }
  if (c.d_model % 128 || c.ffn % 128) { set_error("d_model/ffn must be multiples of 128"); return WSEG_ERR_INVALID; }
  if (c.spec_cols != 2 * c.enc_positions || c.enc_positions > 512 || c.enc_positions < 128) { set_error("spec_cols %d / enc_positions %d unsupported", c.spec_cols, c.enc_positions); return WSEG_ERR_INVALID; }
  if (c.n_mels <= 0 || c.n_mels > 96) { set_error("n_mels %d unsupported", c.n_mels); return WSEG_ERR_INVALID; }
  if (c.dec_positions <= 0 || c.dec_positions > 512) { set_error("dec_positions %d unsupported", c.dec_positions); return WSEG_ERR_INVALID; }
  if (c.dtype < WSEG_F32 || c.dtype > WSEG_F16M6) { set_error("dtype %d unsupported", c.dtype); return WSEG_ERR_INVALID; }
  if (c.enc_layers <= 0 || c.dec_layers <= 0 || c.vocab <= 0) { set_error("bad layer/vocab counts"); return WSEG_ERR_INVALID; }
  return WSEG_OK;
}

#define WSEG_TRY(expr) do { int _s = (expr); if (_s != WSEG_OK) return _s; } while (0)

// WSEG_F16M6: the activation operand of a GEMM (hi | lo rows, row length lda == K logical columns) as M6 rows in `scratch`
int to_mx(const wseg_model* m, const void*& A, int M, int K, char* scratch, hipStream_t s) {
  if (!m->mx) return WSEG_OK;
  if (!scratch) { set_error("M6 operand scratch missing"); return WSEG_ERR_STATE; }
  const int st = launch_x3_to_m6(A, scratch, (size_t)M, K, false, s);
  A = scratch;
  return st;
}

// mxa: scratch for the M6 image of A (WSEG_F16M6; null when A already is M6 rows)
int gemm(const wseg_model* m, EpiKind epi, const void* A, int lda, const void* Wt, int ldw, int M, int N, int K,
         const EpiParams& ep, const DecPlan* p, hipStream_t s, char* mxa = nullptr, int plan_m = 0) {
  if (m->mx && mxa) { const int st = to_mx(m, A, M, K, mxa, s); if (st != WSEG_OK) return st; }
  GemmArgs g;
  g.plan_m = plan_m;
  g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.ep = ep;
  if (p) { g.splitk_ws = (float*)p->splitk; g.splitk_ws_bytes = p->splitk_bytes; }
  return launch_gemm(m->cfg.dtype, epi, g, s);
}

int run_encoder(wseg_model* m, const float* feats, int W, Plan& p, void* enc_out, hipStream_t s) {
  const wseg_model_config& c = m->cfg;
  // dt: dtype of the non-GEMM kernels whose output is NOT a GEMM operand in M6 form; gdt: the model's dtype (WSEG_F16M6: LayerNorm,
  // the attention and the large-tile GEMM epilogues write M6 rows directly; what still arrives as hi | lo rows is converted, mxa)
  const int dt = m->sdt, gdt = c.dtype, d = c.d_model, H = c.n_heads, ffn = c.ffn, T = c.enc_positions, Tp = m->tp;
  const int M1 = W * c.spec_cols, M = W * T;
  const size_t qkv_bytes = (size_t)W * H * Tp * 64 * m->es;
  // pad rows of Q/K and pad columns of V^T must be finite (they are multiplied by exact zeros)
  WSEG_HIP_CHECK(hipMemsetAsync(p.q, 0, qkv_bytes, s));
  WSEG_HIP_CHECK(hipMemsetAsync(p.k, 0, qkv_bytes, s));
  WSEG_HIP_CHECK(hipMemsetAsync(p.vt, 0, qkv_bytes, s));
  WSEG_TRY(launch_im2col_conv1(dt, feats, p.a1, W, c.n_mels, c.spec_cols, m->kp1, s));
  EpiParams e;
  e.bias = m->conv1_b; e.out = p.h1; e.ldc = d;
  WSEG_TRY(gemm(m, EPI_GELU, p.a1, m->kp1, m->conv1_w, m->kp1, M1, d, m->kp1, e, nullptr, s, p.mxa));
  const bool h1_mx = gemm_out_is_mx(gdt, M1, d, m->kp1);       // the im2col of conv2 is a 16-byte copy: a2 inherits h1's row format
  WSEG_TRY(launch_im2col_conv2(dt, p.h1, p.a2, W, c.spec_cols, d, s));
  e = EpiParams();
  e.bias = m->conv2_b; e.out = p.x; e.ldc = d; e.pos = m->enc_pos; e.pos_rows = T;
  WSEG_TRY(gemm(m, EPI_GELU_POS, p.a2, 3 * d, m->conv2_w, 3 * d, M, d, 3 * d, e, nullptr, s, h1_mx ? nullptr : p.mxa));
  const bool attn_mx = enc_attention_writes_mx(gdt), ffn_mx = gemm_out_is_mx(gdt, M, ffn, d);
  for (int l = 0; l < c.enc_layers; ++l) {
    const EncLayer& L = m->enc[l];
    WSEG_TRY(launch_layernorm(gdt, (const float*)p.x, L.ln1_g, L.ln1_b, p.y, M, d, s));
    e = EpiParams();
    e.bias = L.qkv_b; e.q = p.q; e.k = p.k; e.v = p.vt; e.d_model = d; e.t_len = T; e.t_pad = Tp; e.n_heads = H; e.scale = 0.125f;
    if (m->x3) { e.qkv_mode = x3_enc_attention_mode(); e.qkv_plane = (size_t)W * H * Tp * 64; }
    e.vt_tiled = enc_attention_vt_tiled(gdt) ? 1 : 0;
    WSEG_TRY(gemm(m, EPI_QKV_ENC, p.y, d, L.qkv_w, d, M, 3 * d, d, e, nullptr, s));
    WSEG_TRY(launch_enc_attention(gdt, p.q, p.k, p.vt, p.y, W, H, T, Tp, d, s));
    e = EpiParams();
    e.bias = L.o_b; e.out = p.x; e.resid = p.x; e.ldc = d;
    WSEG_TRY(gemm(m, EPI_RESID, p.y, d, L.o_w, d, M, d, d, e, nullptr, s, attn_mx ? nullptr : p.mxa));
    WSEG_TRY(launch_layernorm(gdt, (const float*)p.x, L.ln2_g, L.ln2_b, p.y, M, d, s));
    e = EpiParams();
    e.bias = L.fc1_b; e.out = p.hbuf; e.ldc = ffn;
    WSEG_TRY(gemm(m, EPI_GELU, p.y, d, L.fc1_w, d, M, ffn, d, e, nullptr, s));
    e = EpiParams();
    e.bias = L.fc2_b; e.out = p.x; e.resid = p.x; e.ldc = d;
    WSEG_TRY(gemm(m, EPI_RESID, p.hbuf, ffn, L.fc2_w, ffn, M, d, ffn, e, nullptr, s, ffn_mx ? nullptr : p.mxa));
  }
  // the encoder output stays hi | lo rows (wseg_encode hands it out as fp32); the cross-K/V GEMMs convert it once (mxe)
  WSEG_TRY(launch_layernorm(dt, (const float*)p.x, m->enc_ln_g, m->enc_ln_b, enc_out, M, d, s));
  return WSEG_OK;
}

// The forced prompt positions 0 .. np - 1 of n windows admitted together (device list `slots`), as one pass of n * np rows through the
// decoder layers instead of np steps of every slot: the decoder weights and the windows' cross-attention K / V are read once for them
// (split-precision modes, up to 4 beams; the admission kernel then starts the slots at position np).
struct PromptPass { const int* slots; int n, np; bool logits; int plan_rows; };      // logits: + the LM head on the LAST position's rows -> p.logits [n][vp]
// plan_rows: the rows the pass's GEMMs are PLANNED for (GemmArgs::plan_m) = max(slots x beams, np), a function of the call and not of the
// admission; n * np <= plan_rows.

// One decoder step for all R rows at position *st.pos (want_logits: final LN + LM head), or the prompt pass `pp` of newly admitted windows
// (row i * np + j = window i at position j; no logits: the next token is forced).
int run_decoder_step(wseg_model* m, DecPlan& p, char* mxa, bool want_logits, hipStream_t s, const PromptPass* pp = nullptr) {
  const wseg_model_config& c = m->cfg;
  const int dt = m->sdt, gdt = c.dtype, d = c.d_model, H = c.n_heads, ffn = c.ffn, Tk = c.enc_positions;
  const size_t self_stride = p.kv_layer_stride;
  const size_t cross_stride = (size_t)p.st.W * H * Tk * cross_kv_row_bytes(m->x3 ? x3_cross_kv_format(gdt, p.st.nb) : 0, m->es);
  DecodeState view = p.st;      // the prompt pass seen by the cross-attention kernel: n "slots" of np "beams", none idle
  if (pp) { view.W = pp->n; view.nb = pp->np; view.done = p.zeros; }
  const DecodeState& st = pp ? view : p.st;
  const int R = st.W * st.nb;
  // The pass runs on the GEMM plans of the call's decode step (slots x beams rows) whatever number of windows was admitted: kernel family,
  // tile size and split-K ranges — the summation order of every output element — then depend on the call's slot count only, like the
  // steps themselves, and not on when a window's neighbours finished.
  const int PR = pp ? pp->plan_rows : 0;
  if (pp) {
    if (!m->x3 || x3_cross_kv_format(gdt, p.st.nb) == 0 || R > p.row_cap || R > PR) { set_error("prompt pass: unsupported mode / row count %d", R); return WSEG_ERR_STATE; }
    WSEG_TRY(launch_prompt_embed(m->mx ? WSEG_F32 : dt, p.st, R, pp->np, m->mx ? m->dec_tok_f32 : m->dec_tok, m->dec_pos, p.dx, d, s));
  } else {
    WSEG_TRY(launch_embed(m->mx ? WSEG_F32 : dt, st, m->mx ? m->dec_tok_f32 : m->dec_tok, m->dec_pos, p.dx, d, s));
  }
  EpiParams e;
  // WSEG_F16M6: LayerNorm outputs (dy) and attention outputs (dattn) are M6 rows already; the FFN hidden (dh) is when its GEMM
  // ran on a large-tile kernel (a_mx: the operand needs no conversion)
  const bool dh_mx = gemm_out_is_mx(gdt, pp ? PR : R, ffn, d, p.splitk_bytes);
  auto gemm_resid_ln = [&](const void* A, int K, const void* Wt, const void* bias, const void* g_, const void* b_, bool a_mx) -> int {
    if (!a_mx) WSEG_TRY(to_mx(m, A, R, K, mxa, s));
    GemmArgs g;
    g.A = A; g.lda = K; g.W = Wt; g.ldw = K; g.M = R; g.N = d; g.K = K; g.plan_m = PR;
    g.ep.bias = bias; g.ep.out = p.dx; g.ep.resid = p.dx; g.ep.ldc = d;
    g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
    return launch_gemm_resid_ln(gdt, g, g_, b_, p.dy, s);
  };
  // y = LN1(x) of layer 0; every later LayerNorm is fused into the reduction of the GEMM that precedes it
  WSEG_TRY(launch_layernorm(gdt, (const float*)p.dx, m->dec[0].ln1_g, m->dec[0].ln1_b, p.dy, R, d, s));
  for (int l = 0; l < c.dec_layers; ++l) {
    const DecLayer& L = m->dec[l];
    {   // q|k|v projection: partial sums only when possible; the attention kernel finishes the reduction
      const void* a_op = p.dy;
      GemmArgs g;
      g.A = a_op; g.lda = d; g.W = L.qkv_w; g.ldw = d; g.M = R; g.N = 3 * d; g.K = d; g.plan_m = PR;
      g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
      PartialInfo pi; bool ok = false;
      WSEG_TRY(launch_gemm_partial(gdt, g, &pi, &ok, s));
      if (pp) {
        if (!ok) {      // the step's plan does not split q | k | v: fp32 rows of an un-split GEMM (bias added by the attention kernel)
          if ((size_t)R * 3 * d > (size_t)p.row_cap * m->vp) { set_error("prompt pass: %d rows of q | k | v do not fit the logits buffer", R); return WSEG_ERR_STATE; }
          e = EpiParams();
          e.out_f32 = (float*)p.logits; e.ldc = 3 * d;
          WSEG_TRY(gemm(m, EPI_F32, a_op, d, L.qkv_w, d, R, 3 * d, d, e, &p, s, nullptr, PR));
        }
        WSEG_TRY(launch_prompt_self_attn(gdt, p.st, (const float*)p.logits, ok ? &pi : nullptr, L.qkv_b, p.sk + l * self_stride, p.sv + l * self_stride,
                                         pp->slots, pp->n, pp->np, p.dattn, H, d, 0.125f, s));
      } else if (ok) {
        WSEG_TRY(launch_dec_self_attn(gdt, st, nullptr, p.sk + l * self_stride, p.sv + l * self_stride, p.dattn, H, d, &pi, L.qkv_b, 0.125f, s));
      } else {
        e = EpiParams();
        e.bias = L.qkv_b; e.q = p.dq; e.k = p.sk + l * self_stride; e.v = p.sv + l * self_stride;
        e.d_model = d; e.n_heads = H; e.pos_ptr = st.pos; e.pos_div = st.nb; e.kv_pt = st.kv_pt; e.kv_npg = st.npg; e.idle_ptr = st.done; e.scale = 0.125f;
        WSEG_TRY(gemm(m, EPI_QKV_DEC, a_op, d, L.qkv_w, d, R, 3 * d, d, e, &p, s));
        WSEG_TRY(launch_dec_self_attn(gdt, st, p.dq, p.sk + l * self_stride, p.sv + l * self_stride, p.dattn, H, d, nullptr, nullptr, 0.125f, s));
      }
    }
    WSEG_TRY(gemm_resid_ln(p.dattn, d, L.o_w, L.o_b, L.ln2_g, L.ln2_b, m->mx));                    // x += attn Wo ; y = LN2(x)
    {   // cross-attention query: same scheme
      const void* a_op = p.dy;
      GemmArgs g;
      g.A = a_op; g.lda = d; g.W = L.cq_w; g.ldw = d; g.M = R; g.N = d; g.K = d; g.plan_m = PR;
      g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
      PartialInfo pi; bool ok = false;
      WSEG_TRY(launch_gemm_partial(gdt, g, &pi, &ok, s));
      if (ok) {
        WSEG_TRY(launch_dec_cross_attn(gdt, st, nullptr, p.ck + l * cross_stride, p.cv + l * cross_stride, p.dattn, H, Tk, d, &pi, L.cq_b, 0.125f, s,
                                       pp ? pp->slots : nullptr));
      } else {
        e = EpiParams();
        e.bias = L.cq_b; e.out = p.dq; e.ldc = d; e.scale = 0.125f;
        WSEG_TRY(gemm(m, EPI_SCALE, a_op, d, L.cq_w, d, R, d, d, e, &p, s, nullptr, PR));
        WSEG_TRY(launch_dec_cross_attn(gdt, st, p.dq, p.ck + l * cross_stride, p.cv + l * cross_stride, p.dattn, H, Tk, d, nullptr, nullptr, 0.125f, s,
                                       pp ? pp->slots : nullptr));
      }
    }
    WSEG_TRY(gemm_resid_ln(p.dattn, d, L.co_w, L.co_b, L.ln3_g, L.ln3_b, dec_cross_attn_writes_mx(gdt, st.nb)));      // x += cross Wo ; y = LN3(x)
    e = EpiParams();
    e.bias = L.fc1_b; e.out = p.dh; e.ldc = ffn;
    WSEG_TRY(gemm(m, EPI_GELU, p.dy, d, L.fc1_w, d, R, ffn, d, e, &p, s, nullptr, PR));
    const bool last = l + 1 == c.dec_layers;                                                 // x += fc2 ; y = next LN1 / final LN
    WSEG_TRY(gemm_resid_ln(p.dh, ffn, L.fc2_w, L.fc2_b, last ? m->dec_ln_g : m->dec[l + 1].ln1_g,
                           last ? m->dec_ln_b : m->dec[l + 1].ln1_b, dh_mx));
  }
  if (want_logits && !pp) {
    e = EpiParams();
    e.out_f32 = (float*)p.logits; e.ldc = m->vp;
    WSEG_TRY(gemm(m, EPI_F32, p.dy, d, m->dec_tok, d, R, m->vp, d, e, nullptr, s));
  }
  if (pp && pp->logits) {      // rows i * np + (np - 1) of the final LayerNorm: row stride np operand rows (es bytes per logical element)
    e = EpiParams();
    e.out_f32 = (float*)p.logits; e.ldc = m->vp;
    WSEG_TRY(gemm(m, EPI_F32, p.dy + (size_t)(pp->np - 1) * d * m->es, pp->np * d, m->dec_tok, d, pp->n, m->vp, d, e, nullptr, s, nullptr, PR));
  }
  return WSEG_OK;
}

}  // namespace

extern "C" int wseg_model_create(const wseg_model_config* cfg, wseg_model** out) {
  if (!cfg || !out) { set_error("wseg_model_create: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(check_geometry(*cfg));
  wseg_model* m = new wseg_model();
  m->cfg = *cfg;
  // bytes per element of weights, parameters and activations: 2 in the 16-bit modes; 4 in f32 AND in the split-precision
  // modes, whose tensors are either fp32 or hi | lo pairs of 16-bit words
  m->es = (cfg->dtype == WSEG_BF16 || cfg->dtype == WSEG_F16) ? 2 : 4;
  m->x3 = cfg->dtype == WSEG_BF16X3 || cfg->dtype == WSEG_F16X3 || cfg->dtype == WSEG_F16M6;
  m->mx = cfg->dtype == WSEG_F16M6;
  m->sdt = storage_dtype(cfg->dtype);
  m->ckv_es = m->es;
  m->kp1 = (int)align_up((size_t)3 * cfg->n_mels, 64);
  m->vp = (int)align_up((size_t)cfg->vocab, 128);
  m->tp = (int)align_up((size_t)cfg->enc_positions, 128);
  const size_t d = cfg->d_model, ffn = cfg->ffn;
  add_slot(m, "enc.conv1.w", &m->conv1_w, d * m->kp1);
  add_slot(m, "enc.conv1.b", &m->conv1_b, d);
  add_slot(m, "enc.conv2.w", &m->conv2_w, d * 3 * d);
  add_slot(m, "enc.conv2.b", &m->conv2_b, d);
  add_slot(m, "enc.pos", &m->enc_pos, (size_t)cfg->enc_positions * d);
  add_slot(m, "enc.ln.g", &m->enc_ln_g, d);
  add_slot(m, "enc.ln.b", &m->enc_ln_b, d);
  add_slot(m, "dec.tok", &m->dec_tok, (size_t)m->vp * d);
  if (m->mx) add_slot(m, "dec.tok.f32", &m->dec_tok_f32, (size_t)m->vp * d);
  add_slot(m, "dec.pos", &m->dec_pos, (size_t)cfg->dec_positions * d);
  add_slot(m, "dec.ln.g", &m->dec_ln_g, d);
  add_slot(m, "dec.ln.b", &m->dec_ln_b, d);
  m->enc.resize(cfg->enc_layers);
  m->dec.resize(cfg->dec_layers);
  for (int i = 0; i < cfg->enc_layers; ++i) {
    EncLayer& L = m->enc[i];
    const std::string p = "enc." + std::to_string(i) + ".";
    add_slot(m, p + "ln1.g", &L.ln1_g, d); add_slot(m, p + "ln1.b", &L.ln1_b, d);
    add_slot(m, p + "qkv.w", &L.qkv_w, 3 * d * d); add_slot(m, p + "qkv.b", &L.qkv_b, 3 * d);
    add_slot(m, p + "o.w", &L.o_w, d * d); add_slot(m, p + "o.b", &L.o_b, d);
    add_slot(m, p + "ln2.g", &L.ln2_g, d); add_slot(m, p + "ln2.b", &L.ln2_b, d);
    add_slot(m, p + "fc1.w", &L.fc1_w, ffn * d); add_slot(m, p + "fc1.b", &L.fc1_b, ffn);
    add_slot(m, p + "fc2.w", &L.fc2_w, d * ffn); add_slot(m, p + "fc2.b", &L.fc2_b, d);
  }
  for (int i = 0; i < cfg->dec_layers; ++i) {
    DecLayer& L = m->dec[i];
    const std::string p = "dec." + std::to_string(i) + ".";
    add_slot(m, p + "ln1.g", &L.ln1_g, d); add_slot(m, p + "ln1.b", &L.ln1_b, d);
    add_slot(m, p + "qkv.w", &L.qkv_w, 3 * d * d); add_slot(m, p + "qkv.b", &L.qkv_b, 3 * d);
    add_slot(m, p + "o.w", &L.o_w, d * d); add_slot(m, p + "o.b", &L.o_b, d);
    add_slot(m, p + "ln2.g", &L.ln2_g, d); add_slot(m, p + "ln2.b", &L.ln2_b, d);
    add_slot(m, p + "cq.w", &L.cq_w, d * d); add_slot(m, p + "cq.b", &L.cq_b, d);
    add_slot(m, p + "ckv.w", &L.ckv_w, 2 * d * d); add_slot(m, p + "ckv.b", &L.ckv_b, 2 * d);
    add_slot(m, p + "co.w", &L.co_w, d * d); add_slot(m, p + "co.b", &L.co_b, d);
    add_slot(m, p + "ln3.g", &L.ln3_g, d); add_slot(m, p + "ln3.b", &L.ln3_b, d);
    add_slot(m, p + "fc1.w", &L.fc1_w, ffn * d); add_slot(m, p + "fc1.b", &L.fc1_b, ffn);
    add_slot(m, p + "fc2.w", &L.fc2_w, d * ffn); add_slot(m, p + "fc2.b", &L.fc2_b, d);
  }
  *out = m;
  return WSEG_OK;
}

static void ring_free(PinnedRing& r) {
  for (int i = 0; i < PinnedRing::N; ++i) if (r.ev[i]) (void)hipEventDestroy(r.ev[i]);
  if (r.host) (void)hipHostFree(r.host);
  r = PinnedRing();
}

extern "C" void wseg_model_destroy(wseg_model* m) {
  if (!m) return;
  Sched& sc = m->sched;
  for (hipEvent_t e : sc.ev_pool) (void)hipEventDestroy(e);
  if (sc.step_graph) (void)hipGraphExecDestroy(sc.step_graph);
  if (sc.cap_stream) (void)hipStreamDestroy(sc.cap_stream);
  ring_free(sc.ring_h2d);
  ring_free(sc.ring_status);
  delete m;
}

extern "C" int wseg_model_set_tensor(wseg_model* m, const char* name, const void* dev_ptr, size_t bytes) {
  if (!m || !name || !dev_ptr) { set_error("wseg_model_set_tensor: null argument"); return WSEG_ERR_INVALID; }
  auto it = m->slots.find(name);
  if (it == m->slots.end()) { set_error("unknown tensor '%s'", name); return WSEG_ERR_INVALID; }
  if (it->second.bytes != bytes) { set_error("tensor '%s': expected %zu bytes, got %zu", name, it->second.bytes, bytes); return WSEG_ERR_INVALID; }
  if (((uintptr_t)dev_ptr) & 15) { set_error("tensor '%s' is not 16-byte aligned", name); return WSEG_ERR_INVALID; }
  *it->second.field = dev_ptr;
  it->second.set = true;
  m->sched.step_graph_key.clear();      // a captured decode step holds the old pointer: recapture on the next call
  return WSEG_OK;
}

extern "C" int wseg_model_ready(const wseg_model* m) {
  if (!m) { set_error("wseg_model_ready: null model"); return WSEG_ERR_INVALID; }
  for (const auto& kv : m->slots)
    if (!kv.second.set) { set_error("tensor '%s' has not been attached", kv.first.c_str()); return WSEG_ERR_STATE; }
  return WSEG_OK;
}

extern "C" size_t wseg_workspace_bytes_kv(const wseg_model* m, int32_t max_windows, int32_t num_beams, int32_t max_length,
                                          int32_t kv_positions_per_slot) {
  if (!m || max_windows <= 0 || num_beams <= 0 || num_beams > MAX_BEAMS || max_length <= 0 || kv_positions_per_slot < 0) return 0;
  Plan p;
  make_plan(m, max_windows, num_beams, max_length,
            kv_units_for(max_windows, max_length, kv_positions_per_slot == 0 ? KV_DEFAULT_POSITIONS : kv_positions_per_slot), nullptr, p);
  return p.total + 256;
}

extern "C" size_t wseg_workspace_bytes(const wseg_model* m, int32_t max_windows, int32_t num_beams, int32_t max_length) {
  return wseg_workspace_bytes_kv(m, max_windows, num_beams, max_length, 0);
}

static char* aligned_base(void* ws) { return (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255); }

extern "C" int wseg_encode(wseg_model* m, const float* feats, int32_t n_windows, void* workspace, size_t workspace_bytes,
                           void* enc_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !feats || !workspace || !enc_out) { set_error("wseg_encode: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(wseg_model_ready(m));
  if (n_windows <= 0) return WSEG_OK;
  Plan p;
  make_plan(m, n_windows, 1, 8, kv_default_units(n_windows, 8), aligned_base(workspace), p);
  if (p.total + 256 > workspace_bytes) { set_error("workspace too small: need %zu, have %zu", p.total + 256, workspace_bytes); return WSEG_ERR_STATE; }
  const size_t feat_stride = (size_t)m->cfg.n_mels * m->cfg.spec_cols;
  for (int w0 = 0; w0 < n_windows; w0 += ENC_CHUNK) {      // passes of at most ENC_CHUNK windows (the encoder buffers' size)
    const int n = std::min(ENC_CHUNK, n_windows - w0);
    WSEG_TRY(run_encoder(m, feats + (size_t)w0 * feat_stride, n, p, p.enc_out, s));
    const size_t rows = (size_t)n * m->cfg.enc_positions, row0 = (size_t)w0 * m->cfg.enc_positions;
    if (m->x3) WSEG_TRY(launch_operand_to_f32(m->sdt, p.enc_out, (float*)enc_out + row0 * m->cfg.d_model, rows, m->cfg.d_model, s));
    else WSEG_HIP_CHECK(hipMemcpyAsync((char*)enc_out + row0 * m->cfg.d_model * m->es, p.enc_out, rows * m->cfg.d_model * m->es, hipMemcpyDeviceToDevice, s));
  }
  return WSEG_OK;
}

// ---- scheduler plumbing -------------------------------------------------------------------------
static int ring_prepare(PinnedRing& r, int cap) {
  if (r.cap >= cap) return WSEG_OK;
  // every entry must be idle before the buffer is replaced
  for (int i = 0; i < PinnedRing::N; ++i) if (r.used[i]) { WSEG_HIP_CHECK(hipEventSynchronize(r.ev[i])); r.used[i] = false; }
  if (r.host) WSEG_HIP_CHECK(hipHostFree(r.host));
  r.host = nullptr;
  WSEG_HIP_CHECK(hipHostMalloc((void**)&r.host, (size_t)PinnedRing::N * cap * sizeof(int), hipHostMallocDefault));
  r.cap = cap;
  for (int i = 0; i < PinnedRing::N; ++i) if (!r.ev[i]) WSEG_HIP_CHECK(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
  return WSEG_OK;
}
// next idle entry (waits for the copy that last used it)
static int ring_acquire(PinnedRing& r, int* idx) {
  const int i = r.next;
  r.next = (r.next + 1) % PinnedRing::N;
  if (r.used[i]) { WSEG_HIP_CHECK(hipEventSynchronize(r.ev[i])); r.used[i] = false; }
  *idx = i;
  return WSEG_OK;
}
static int h2d_list(Sched& ln, const int* vals, int n, int* dev, hipStream_t s) {
  int i;
  WSEG_TRY(ring_acquire(ln.ring_h2d, &i));
  int* h = ln.ring_h2d.host + (size_t)i * ln.ring_h2d.cap;
  memcpy(h, vals, (size_t)n * sizeof(int));
  WSEG_HIP_CHECK(hipMemcpyAsync(dev, h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
  WSEG_HIP_CHECK(hipEventRecord(ln.ring_h2d.ev[i], s));
  ln.ring_h2d.used[i] = true;
  return WSEG_OK;
}
static int timing_event(Sched& ln, hipStream_t s, int* idx) {
  if (ln.ev_used == ln.ev_pool.size()) { hipEvent_t e; WSEG_HIP_CHECK(hipEventCreate(&e)); ln.ev_pool.push_back(e); }
  *idx = (int)ln.ev_used++;
  WSEG_HIP_CHECK(hipEventRecord(ln.ev_pool[*idx], s));
  return WSEG_OK;
}

// Decode of n_windows windows through S window slots on stream s.
//
// The reference decodes batch by batch (model.py:653): a batch runs until its slowest window has finished.  Here a
// finished window's slot is retired and handed to the next queued window while the other slots keep decoding: every
// slot has its own position, every per-step kernel skips idle slots (so they cost no K/V traffic), and the captured
// step graph never changes.  Windows are independent, so the tokens of a window do not depend on which slot it ran in
// or on what ran beside it (row-independent kernels, fixed row count => fixed tile / split-K plan).
// The host runs at most `lookahead` steps ahead of the device: the per-step status mirror (done flag of every slot)
// is read behind an event, which both bounds the wasted steps after the last window finishes and tells the scheduler
// which slots to retire / refill.
//
// Self-attention K / V are PAGED (wseg_kernels.h): the host knows the position every occupied slot feeds at step t (t minus the
// step it was admitted at), so it hands out a pool unit whenever a slot crosses a page boundary — before the step is launched,
// through a small page-table update kernel — and takes the slot's units back when it retires.  The pool is sized for the
// expected length; when it runs short the YOUNGEST slot is preempted (aborted on the device, its window re-queued and later
// decoded again from scratch: the same tokens), so the oldest window always makes progress, and admissions keep a one-page
// cushion per window in flight and pause after a preemption until a slot retires.
static int generate_windows(wseg_model* m, const float* feats, int n_windows, const wseg_generate_params* gp, char* base, int S,
                            int kv_units, int32_t* out_tokens, int32_t* out_lengths, hipStream_t s) {
  Sched& ln = m->sched;
  const wseg_model_config& c = m->cfg;
  const int nb = gp->num_beams, P = gp->prompt_len, L = gp->max_length;
  const int G = gp->refill_min > 0 ? gp->refill_min : (S >= 16 ? S / 8 : 1);                // admit once this many slots are free
  const int K = gp->lookahead > 0 ? (gp->lookahead < PinnedRing::N - 2 ? gp->lookahead : PinnedRing::N - 2) : 1;
  Plan p;
  make_plan(m, S, nb, L, kv_units, base, p);
  DecPlan& q = p.dec;
  DecodeState& st = q.st;
  st.P = P; st.eos = gp->eos_token_id; st.pad = gp->pad_token_id; st.max_length = L; st.length_penalty = gp->length_penalty;
  for (int i = 0; i < 8; ++i) st.prompt[i] = i < P ? gp->prompt[i] : 0;
  st.win_max_length = gp->window_max_length;
  st.top_k = (nb == 1 && gp->top_k > 1) ? gp->top_k : 1;
  st.top_p = gp->top_p;
  st.seed = (const unsigned long long*)q.seed_dev;
  WSEG_TRY(ring_prepare(ln.ring_h2d, 2 * S > 2 ? 2 * S : 2));
  WSEG_TRY(ring_prepare(ln.ring_status, S));
  wseg_generate_stats& stats = m->stats;
  stats = wseg_generate_stats();
  stats.n_windows = n_windows; stats.n_slots = S; stats.kv_units_total = kv_units;
  {   // the seed lives in device memory (read by the sampling kernel): per-call values do not invalidate the step graph
    const unsigned long long sd = gp->seed;
    int words[2];
    memcpy(words, &sd, 8);
    WSEG_TRY(h2d_list(ln, words, 2, q.seed_dev, s));
  }
  WSEG_TRY(launch_build_suppress_mask((unsigned char*)q.mask, c.vocab, gp->suppress_tokens, gp->n_suppress,
                                      gp->begin_suppress_tokens, gp->n_begin_suppress, s));
  WSEG_TRY(launch_decode_reset(st, s));
  WSEG_HIP_CHECK(hipMemsetAsync(q.zeros, 0, (size_t)S * sizeof(int), s));

  const int d = c.d_model, H = c.n_heads, Tk = c.enc_positions;
  const int kv24 = m->x3 ? x3_cross_kv_format(c.dtype, nb) : 0;
  // prompt pass: the first NPF forced positions of every admission run as one pass (run_decoder_step, PromptPass) and the slots start
  // at position NPF.  Split-precision modes up to 4 beams (24-bit / block-floating-point cross K / V); the f32 and plain 16-bit modes step through the prompt.
  const bool prompt_pass = getenv("WSEG_NO_PROMPT_PASS") == nullptr;      // test knob (read per call): step through the prompt instead
  // ... and when the whole prompt fits the pass (P <= 4) and a step follows anyway (L >= P + 2), the pass also runs position P - 1 — the FIRST
  // GENERATED step, whose beams are still copies: one row per window through the layers and the LM head instead of a full decode step of every
  // beam row (which would stream the windows' cross K / V once more) — and the admission finishes that step's bookkeeping for its slots
  // (candidates from the window's one logits row, beam / greedy step on the admitted list).  NPF: positions the pass covers; POS0: the
  // position the slot is at when the decode loop first steps it.
  const bool pass_ok = prompt_pass && m->x3 && kv24 != 0;
  const bool merged = pass_ok && P <= 4 && L >= P + 2 && getenv("WSEG_NO_FIRST_STEP_MERGE") == nullptr;
  const int NPF = !pass_ok ? 0 : (merged ? P : std::min(P - 1, 4));
  const int POS0 = merged ? P : NPF;
  const size_t cross_stride = (size_t)S * H * Tk * cross_kv_row_bytes(kv24, m->es);
  const size_t feat_stride = (size_t)c.n_mels * c.spec_cols;
  const int npg = st.npg;

  // host view of the slots
  std::vector<int> slot_win(S, -1), slot_from(S, 0);   // window in the slot (-1 = free), first step whose status counts for it (at which
                                                       // the slot is at position NPF)
  std::vector<std::vector<int>> slot_units(S);         // pool units the slot holds, in page order
  std::vector<int> free_slots, free_units, tmp_a, tmp_b;
  for (int i = S - 1; i >= 0; --i) free_slots.push_back(i);   // popped from the back: lowest slot first
  for (int i = kv_units - 1; i >= 0; --i) free_units.push_back(i);
  std::deque<int> queue;                               // windows waiting for a slot (preempted windows return to the front)
  for (int i = 0; i < n_windows; ++i) queue.push_back(i);
  int in_flight = 0, t = 0, units_in_use = 0;
  bool hold_admission = false;                         // set by a preemption, cleared by the next retirement
  bool first_admission = true;
  bool snap_ok = false;                                // did every window of the call start together (first-logits snapshot)?
  m->first_logits_valid = false;

  // encoder + cross-K/V of the consecutive windows [w0, w0 + n) into the slots listed at q.adm_slots + off (device)
  auto encode_run = [&](int w0, int n, int off) -> int {
    for (int c0 = 0; c0 < n; c0 += ENC_CHUNK) {
      const int nc = std::min(ENC_CHUNK, n - c0), wc = w0 + c0;
      int e0, e1, e2;
      WSEG_TRY(timing_event(ln, s, &e0));
      const char* enc_rows = p.enc_out;
      if (gp->encoder_output && m->x3)      // handed over as fp32: re-split into operand rows for the cross-K/V GEMMs
        WSEG_TRY(launch_f32_to_operand(m->sdt, (const float*)gp->encoder_output + (size_t)wc * Tk * d, p.enc_out, (size_t)nc * Tk, d, s));
      else if (gp->encoder_output) enc_rows = (const char*)gp->encoder_output + (size_t)wc * Tk * d * m->es;
      else WSEG_TRY(run_encoder(m, feats + (size_t)wc * feat_stride, nc, p, p.enc_out, s));
      WSEG_TRY(timing_event(ln, s, &e1));
      if (m->mx) {      // M6 rows of the encoder output ONCE for the cross-K/V GEMMs of all decoder layers
        const void* er = enc_rows;
        WSEG_TRY(to_mx(m, er, nc * Tk, d, p.mxe, s));
        enc_rows = (const char*)er;
      }
      for (int l = 0; l < c.dec_layers; ++l) {     // cross-attention K/V of every decoder layer, once per window (shared by its beams)
        EpiParams e;
        e.bias = m->dec[l].ckv_b; e.k = q.ck + l * cross_stride; e.v = q.cv + l * cross_stride;
        e.d_model = d; e.t_len = Tk; e.n_heads = H; e.slot_map = q.adm_slots + off + c0; e.kv24 = kv24;
        // planned for a full encoder chunk whatever the admission holds: below ~5 windows the launcher would otherwise pick the stream
        // family with a split-K count that follows the row count — a window's cross K / V (and through their block-floating-point
        // rounding its tokens on a near-tie) would depend on how many windows were admitted with it.  The un-split large-tile
        // families compute every output element as the same K-ordered MFMA chain (r06, with GemmArgs::plan_m for the pass).
        WSEG_TRY(gemm(m, EPI_KV_CROSS, enc_rows, d, m->dec[l].ckv_w, d, nc * Tk, 2 * d, d, e, &q, s, nullptr, ENC_CHUNK * Tk));
      }
      WSEG_TRY(timing_event(ln, s, &e2));
      ln.ev_enc.push_back(e0); ln.ev_enc.push_back(e1); ln.ev_ckv.push_back(e1); ln.ev_ckv.push_back(e2);
    }
    return WSEG_OK;
  };
  // the first n windows of the queue into n free slots; their decode state starts at position 0
  auto admit = [&](int n) -> int {
    tmp_a.clear(); tmp_b.clear();
    for (int i = 0; i < n; ++i) {
      const int sl = free_slots.back(); free_slots.pop_back();
      const int w = queue.front(); queue.pop_front();
      tmp_a.push_back(sl); tmp_b.push_back(w);
      slot_win[sl] = w; slot_from[sl] = t;
    }
    WSEG_TRY(h2d_list(ln, tmp_a.data(), n, q.adm_slots, s));
    WSEG_TRY(h2d_list(ln, tmp_b.data(), n, q.adm_wins, s));
    for (int i = 0; i < n;) {                      // runs of consecutive window indices (a re-queued window breaks a run)
      int j = i + 1;
      while (j < n && tmp_b[j] == tmp_b[j - 1] + 1) ++j;
      WSEG_TRY(encode_run(tmp_b[i], j - i, i));
      i = j;
    }
    WSEG_TRY(launch_decode_admit(st, q.adm_slots, q.adm_wins, n, NPF, merged ? P - 1 : NPF, s));
    if (NPF > 0) {
      // the first page of every admitted slot now (the refill rule left a pool unit for each), then the prompt pass in chunks that fit
      // the decode step's row buffers
      tmp_b.clear();
      for (int i = 0; i < n; ++i) {
        const int sl = tmp_a[i];
        if (free_units.empty()) { set_error("admission without a pool unit per window"); return WSEG_ERR_STATE; }
        const int u = free_units.back(); free_units.pop_back();
        slot_units[sl].push_back(u);
        ++units_in_use;
        tmp_b.push_back(sl * npg); tmp_b.push_back(u);
      }
      if (units_in_use > stats.kv_units_peak) stats.kv_units_peak = units_in_use;
      WSEG_TRY(h2d_list(ln, tmp_b.data(), (int)tmp_b.size(), q.kv_pairs, s));
      WSEG_TRY(launch_kv_assign(q.kv_pt, q.kv_pairs, n, s));
      const int plan_rows = std::max(S * nb, NPF), chunk = std::max(1, std::min(q.row_cap, plan_rows) / NPF);
      for (int c0 = 0; c0 < n; c0 += chunk) {
        const int nc = std::min(chunk, n - c0);
        const PromptPass pp = {q.adm_slots + c0, nc, NPF, merged, plan_rows};
        WSEG_TRY(run_decoder_step(m, q, p.mxa, false, s, &pp));
        if (merged) {
          if (snap_ok && stats.n_admissions == 0) {      // wseg_debug_first_logits (all windows of the call start together: window i sits in slot i): every beam row of
            for (int j = 0; j < nb; ++j)      // a window gets the window's row
              WSEG_HIP_CHECK(hipMemcpy2DAsync(q.first_logits + ((size_t)c0 * nb + j) * m->vp * 4, (size_t)nb * m->vp * 4, q.logits, (size_t)m->vp * 4,
                                              (size_t)m->vp * 4, (size_t)nc, hipMemcpyDeviceToDevice, s));
            if (c0 + nc == n) m->first_logits_valid = true;
          }
          WSEG_TRY(launch_row_topk(st, (const float*)q.logits, (float*)q.tk_val, (int*)q.tk_idx, (float*)q.tk_stat, s, q.adm_slots + c0, nc));
          if (nb == 1) WSEG_TRY(launch_greedy_step(st, s, q.adm_slots + c0, nc));
          else WSEG_TRY(launch_beam_step(st, s, q.adm_slots + c0, nc));
        }
      }
    }
    in_flight += n;
    stats.n_admissions += 1;
    return WSEG_OK;
  };
  // take a slot out of flight without output (its window goes back to the head of the queue)
  auto preempt = [&](int sl) -> int {
    WSEG_TRY(h2d_list(ln, &sl, 1, q.ret_slots, s));
    WSEG_TRY(launch_decode_abort(st, q.ret_slots, 1, s));
    queue.push_front(slot_win[sl]);
    slot_win[sl] = -1;
    for (int u : slot_units[sl]) free_units.push_back(u);
    units_in_use -= (int)slot_units[sl].size();
    slot_units[sl].clear();
    free_slots.push_back(sl);
    std::sort(free_slots.begin(), free_slots.end(), [](int a, int b) { return a > b; });
    --in_flight;
    stats.n_preemptions += 1;
    hold_admission = true;
    return WSEG_OK;
  };
  // pool units for every occupied slot that enters a new page at step t (position t - slot_from: the host's upper bound — a slot
  // that finished inside the look-ahead window is idle on the device and simply does not use the page)
  auto assign_pages = [&]() -> int {
    tmp_a.clear();
    for (int sl = 0; sl < S; ++sl) {
      if (slot_win[sl] < 0) continue;
      const int pos = t - slot_from[sl] + POS0;
      if (pos >= L || pos % KV_PAGE) continue;
      while (free_units.empty()) {
        int victim = -1;                                // youngest slot that holds pages (never the requester)
        for (int v = 0; v < S; ++v)
          if (v != sl && slot_win[v] >= 0 && !slot_units[v].empty() && (victim < 0 || slot_from[v] >= slot_from[victim])) victim = v;
        if (victim < 0) { set_error("self-attention K/V pool exhausted by one window (pool of %d units)", kv_units); return WSEG_ERR_STATE; }
        // an assignment already queued for the victim in this pass is void: drop it
        for (size_t i = 0; i + 1 < tmp_a.size();) { if (tmp_a[i] / npg == victim) tmp_a.erase(tmp_a.begin() + i, tmp_a.begin() + i + 2); else i += 2; }
        WSEG_TRY(preempt(victim));
      }
      const int u = free_units.back(); free_units.pop_back();
      slot_units[sl].push_back(u);
      ++units_in_use;
      tmp_a.push_back(sl * npg + pos / KV_PAGE); tmp_a.push_back(u);
    }
    if (units_in_use > stats.kv_units_peak) stats.kv_units_peak = units_in_use;
    if (!tmp_a.empty()) {
      WSEG_TRY(h2d_list(ln, tmp_a.data(), (int)tmp_a.size(), q.kv_pairs, s));
      WSEG_TRY(launch_kv_assign(q.kv_pt, q.kv_pairs, (int)tmp_a.size() / 2, s));
    }
    return WSEG_OK;
  };

  // One decode step of every active slot: decoder layers, LM head, candidates, bookkeeping (which also advances the slot).
  auto enqueue_step = [&](bool snapshot_logits, hipStream_t qs) -> int {
    WSEG_TRY(run_decoder_step(m, q, p.mxa, true, qs));
    if (snapshot_logits)
      WSEG_HIP_CHECK(hipMemcpyAsync(q.first_logits, q.logits, (size_t)S * nb * m->vp * 4, hipMemcpyDeviceToDevice, qs));
    WSEG_TRY(launch_row_topk(st, (const float*)q.logits, (float*)q.tk_val, (int*)q.tk_idx, (float*)q.tk_stat, qs));
    if (nb == 1) WSEG_TRY(launch_greedy_step(st, qs));
    else WSEG_TRY(launch_beam_step(st, qs));
    return WSEG_OK;
  };
  // The step reads every step-dependent value (positions, tokens, ancestry, idle flags, the page table, the sampling seed) from
  // device memory, so ONE captured graph serves all steps and later calls: replay costs ~1.6 us per kernel instead of ~5 us per
  // eager launch.  The key holds what the captured kernels take BY VALUE; per-window length caps are applied by the
  // admission kernel (launched outside the graph) and the weight pointers invalidate the key in wseg_model_set_tensor.
  static const bool use_graph = getenv("WSEG_NO_GRAPH") == nullptr;
  std::vector<unsigned char> key;
  {
    auto put = [&](const void* ptr, size_t n) { const unsigned char* b = (const unsigned char*)ptr; key.insert(key.end(), b, b + n); };
    put(&base, sizeof(base)); put(&S, 4); put(&nb, 4); put(&L, 4); put(&kv_units, 4);
    put(&st.P, 4); put(&st.eos, 4); put(&st.pad, 4); put(&st.length_penalty, 4); put(st.prompt, sizeof(st.prompt));
    put(&st.top_k, 4); put(&st.top_p, 4);
  }
  auto launch_step = [&]() -> int {
    // the first generated step of a call whose windows all start together is launched eagerly with the logits snapshot
    // (wseg_debug_first_logits); every other step replays the graph
    const bool snap = !merged && t == P - 1 - POS0 && snap_ok && stats.n_preemptions == 0;
    if (snap) m->first_logits_valid = true;
    if (snap || !use_graph) return enqueue_step(snap, s);
    if (!ln.step_graph || ln.step_graph_key != key) {
      if (ln.step_graph) { (void)hipGraphExecDestroy(ln.step_graph); ln.step_graph = nullptr; }
      hipGraph_t graph = nullptr;
      if (!ln.cap_stream) WSEG_HIP_CHECK(hipStreamCreateWithFlags(&ln.cap_stream, hipStreamNonBlocking));
      WSEG_HIP_CHECK(hipStreamBeginCapture(ln.cap_stream, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue_step(false, ln.cap_stream);
      const hipError_t ec = hipStreamEndCapture(ln.cap_stream, &graph);
      if (rc != WSEG_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (ec != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ec)); return WSEG_ERR_HIP; }
      WSEG_HIP_CHECK(hipGraphInstantiate(&ln.step_graph, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      ln.step_graph_key = key;
    }
    WSEG_HIP_CHECK(hipGraphLaunch(ln.step_graph, s));
    return WSEG_OK;
  };

  // status mirror of step u lives in ring entry status_idx[u % N]
  int status_idx[PinnedRing::N];
  bool step_queued[PinnedRing::N];             // were windows still waiting in the queue when the step was launched?
  auto consume_status = [&](int u) -> int {      // retire every slot that step u left finished
    const int ri = status_idx[u % PinnedRing::N];
    WSEG_HIP_CHECK(hipEventSynchronize(ln.ring_status.ev[ri]));
    ln.ring_status.used[ri] = false;
    const int* done = ln.ring_status.host + (size_t)ri * ln.ring_status.cap;
    tmp_a.clear();
    int active = 0;
    for (int sl = 0; sl < S; ++sl) {
      if (slot_win[sl] < 0 || u < slot_from[sl]) continue;
      if (done[sl]) { tmp_a.push_back(sl); slot_win[sl] = -1; }
      else ++active;
    }
    stats.slot_steps_active += active + (int64_t)tmp_a.size();
    if (step_queued[u % PinnedRing::N]) {
      stats.queued_slot_steps_active += active + (int64_t)tmp_a.size();
      stats.queued_slot_steps_total += S;
    }
    if (!tmp_a.empty()) {
      const int n = (int)tmp_a.size();
      WSEG_TRY(h2d_list(ln, tmp_a.data(), n, q.ret_slots, s));
      WSEG_TRY(launch_finalize(st, q.ret_slots, n, out_tokens, out_lengths, s));
      for (int sl : tmp_a) {
        free_slots.push_back(sl);
        for (int v : slot_units[sl]) free_units.push_back(v);
        units_in_use -= (int)slot_units[sl].size();
        slot_units[sl].clear();
      }
      std::sort(free_slots.begin(), free_slots.end(), [](int a, int b) { return a > b; });
      in_flight -= n;
      hold_admission = false;
    }
    return WSEG_OK;
  };

  int consumed = 0;                               // statuses of steps [0, consumed) have been processed
  while (true) {
    const int rem = (int)queue.size();
    if (rem > 0 && !(hold_admission && in_flight > 0)) {
      // refill rule: enough free slots, or the rest of the queue, or nothing else is running — and a pool unit for every
      // admitted window on top of one spare unit per window in flight (16+ steps without a preemption)
      int n_adm = std::min(rem, (int)free_slots.size());
      const int spare = (int)free_units.size() - in_flight;
      n_adm = std::min(n_adm, in_flight == 0 ? (int)free_units.size() : std::max(spare, 0));
      if (n_adm > 0 && (n_adm >= G || n_adm == rem || in_flight == 0)) {
        if (first_admission) snap_ok = n_adm == n_windows;
        first_admission = false;
        WSEG_TRY(admit(n_adm));
      }
    }
    const bool drained = queue.empty();
    if (in_flight == 0) {
      if (!drained) { set_error("scheduler stalled with %d windows queued", (int)queue.size()); return WSEG_ERR_STATE; }
      break;
    }
    if (drained) {                // nothing left to admit later: stop launching once every window in flight must have ended
      bool may_run = false;       // (a window admitted before step f feeds its last position, L - 2, at step f + L - 2 - POS0)
      for (int sl = 0; sl < S && !may_run; ++sl) may_run = slot_win[sl] >= 0 && t < slot_from[sl] + L - 1 - POS0;
      if (!may_run) break;
    }
    WSEG_TRY(assign_pages());
    WSEG_TRY(launch_step());
    {   // mirror the idle flags of this step
      int ri;
      WSEG_TRY(ring_acquire(ln.ring_status, &ri));
      WSEG_HIP_CHECK(hipMemcpyAsync(ln.ring_status.host + (size_t)ri * ln.ring_status.cap, st.done, (size_t)S * sizeof(int),
                                    hipMemcpyDeviceToHost, s));
      WSEG_HIP_CHECK(hipEventRecord(ln.ring_status.ev[ri], s));
      ln.ring_status.used[ri] = true;
      status_idx[t % PinnedRing::N] = ri;
      step_queued[t % PinnedRing::N] = !drained;
    }
    ++t;
    stats.slot_steps_total += S;
    // stay at most K steps ahead of the device
    while (consumed < t - K) WSEG_TRY(consume_status(consumed++));
  }
  while (consumed < t) WSEG_TRY(consume_status(consumed++));
  if (in_flight != 0 || !queue.empty()) { set_error("scheduler ended with %d windows in flight, %d queued", in_flight, (int)queue.size()); return WSEG_ERR_STATE; }
  stats.n_steps = t;
  return WSEG_OK;
}

// Decode of n_windows windows through n_slots window slots (one decode loop on the caller's stream).  More concurrency comes
// from more SLOTS: independent slot groups on separate streams ("lanes", round 2) measured exactly as one group with their
// total slot count (profiles/README.md r02) and were removed in round 3.
extern "C" int wseg_generate(wseg_model* m, const float* feats, int32_t n_windows, const wseg_generate_params* gp,
                             void* workspace, size_t workspace_bytes, int32_t* out_tokens, int32_t* out_lengths,
                             void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !gp || (!feats && !gp->encoder_output) || !workspace || !out_tokens || !out_lengths) { set_error("wseg_generate: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(wseg_model_ready(m));
  if (n_windows <= 0) return WSEG_OK;
  const wseg_model_config& c = m->cfg;
  const int nb = gp->num_beams, P = gp->prompt_len, L = gp->max_length;
  if (nb < 1 || nb > MAX_BEAMS) { set_error("num_beams %d unsupported (1..%d)", nb, MAX_BEAMS); return WSEG_ERR_INVALID; }
  if (P < 1 || P > 8 || L <= P || L > c.dec_positions) { set_error("prompt_len %d / max_length %d unsupported", P, L); return WSEG_ERR_INVALID; }
  if (gp->n_suppress < 0 || gp->n_begin_suppress < 0 || (gp->n_suppress && !gp->suppress_tokens) || (gp->n_begin_suppress && !gp->begin_suppress_tokens)) {
    set_error("bad suppress-token lists"); return WSEG_ERR_INVALID;
  }
  if (gp->n_slots < 0 || gp->refill_min < 0 || gp->lookahead < 0) { set_error("bad scheduler parameters"); return WSEG_ERR_INVALID; }
  if (nb == 1 && gp->top_k > MAX_CAND) { set_error("top_k %d unsupported (sampling draws among at most %d candidates)", gp->top_k, MAX_CAND); return WSEG_ERR_INVALID; }
  const int S = gp->n_slots > 0 && gp->n_slots < n_windows ? gp->n_slots : n_windows;       // window slots
  // self-attention K / V pool: every unit the workspace has room for behind the fixed buffers, at most a full set (every slot
  // to max_length), at least one slot's worth (a lone window can always finish: the scheduler's progress guarantee)
  const int npg = kv_pages(L), full = S * npg;
  Plan p;
  make_plan(m, S, nb, L, 0, nullptr, p);
  const size_t fixed = p.total + 256;
  long units = workspace_bytes > fixed ? (long)((workspace_bytes - fixed) / kv_unit_bytes(m, nb)) : 0;
  if (units > full) units = full;
  while (units >= npg) {      // the per-layer pools are aligned: make sure the chosen count really fits
    make_plan(m, S, nb, L, (int)units, nullptr, p);
    if (p.total + 256 <= workspace_bytes) break;
    --units;
  }
  if (units < npg) {
    make_plan(m, S, nb, L, npg, nullptr, p);
    set_error("workspace too small: need at least %zu bytes (wseg_workspace_bytes: %zu), have %zu", p.total + 256,
              wseg_workspace_bytes(m, S, nb, L), workspace_bytes);
    return WSEG_ERR_STATE;
  }
  Sched& sc = m->sched;
  sc.ev_used = 0; sc.ev_enc.clear(); sc.ev_ckv.clear();
  m->timing_valid = false;
  WSEG_TRY(timing_event(sc, s, &m->ev_total[0]));
  WSEG_TRY(generate_windows(m, feats, n_windows, gp, aligned_base(workspace), S, (int)units, out_tokens, out_lengths, s));
  WSEG_TRY(timing_event(sc, s, &m->ev_total[1]));
  m->last_W = S; m->last_nb = nb; m->last_L = L; m->last_units = (int)units;
  m->timing_valid = true;
  return WSEG_OK;
}

extern "C" int wseg_debug_first_logits(wseg_model* m, void* workspace, float* out, int32_t n_rows, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !workspace || !out || n_rows <= 0) { set_error("wseg_debug_first_logits: bad argument"); return WSEG_ERR_INVALID; }
  if (m->last_W <= 0 || n_rows > m->last_W * m->last_nb || !m->first_logits_valid) {
    set_error("no matching wseg_generate call (all windows must have started together)"); return WSEG_ERR_STATE;
  }
  Plan p;
  make_plan(m, m->last_W, m->last_nb, m->last_L, m->last_units, aligned_base(workspace), p);
  WSEG_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)m->cfg.vocab * 4, p.dec.first_logits, (size_t)m->vp * 4,
                                  (size_t)m->cfg.vocab * 4, (size_t)n_rows, hipMemcpyDeviceToDevice, s));
  return WSEG_OK;
}

extern "C" int wseg_last_timing(const wseg_model* m, float out[4]) {
  if (!m || !out) { set_error("wseg_last_timing: null argument"); return WSEG_ERR_INVALID; }
  if (!m->timing_valid) { set_error("no completed wseg_generate call to time"); return WSEG_ERR_STATE; }
  const Sched& sc = m->sched;
  WSEG_HIP_CHECK(hipEventSynchronize(sc.ev_pool[m->ev_total[1]]));
  auto sum_pairs = [&](const std::vector<int>& v, float* acc) -> int {
    *acc = 0.f;
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      float ms = 0.f;
      WSEG_HIP_CHECK(hipEventElapsedTime(&ms, sc.ev_pool[v[i]], sc.ev_pool[v[i + 1]]));
      *acc += ms;
    }
    return WSEG_OK;
  };
  float total = 0.f;
  WSEG_TRY(sum_pairs(sc.ev_enc, &out[0]));
  WSEG_TRY(sum_pairs(sc.ev_ckv, &out[1]));
  WSEG_HIP_CHECK(hipEventElapsedTime(&total, sc.ev_pool[m->ev_total[0]], sc.ev_pool[m->ev_total[1]]));
  out[2] = total - out[0] - out[1];
  out[3] = (float)m->stats.n_steps;
  return WSEG_OK;
}

extern "C" int wseg_last_stats(const wseg_model* m, wseg_generate_stats* out) {
  if (!m || !out) { set_error("wseg_last_stats: null argument"); return WSEG_ERR_INVALID; }
  if (!m->timing_valid) { set_error("no completed wseg_generate call"); return WSEG_ERR_STATE; }
  *out = m->stats;
  return WSEG_OK;
}
