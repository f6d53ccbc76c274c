// Model object, workspace planning and the encode / generate drivers of libwseg.
//
// Replaces HF WhisperForConditionalGeneration as the reference drives it (reference model.py:626-676):
// encoder (HF modeling_whisper.py:592-646), decoder with KV cache (:690-796), tied LM head (:1080) and
// greedy / beam-search decoding (HF generation/utils.py:3208-3510).  The host code below only enqueues
// kernels on the caller's stream; all decoding state lives in the caller-provided workspace.
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

struct DecPlan {   // decoder-side buffers of one cohort of windows
  int W = 0, w0 = 0;           // windows in the cohort, index of its first window
  char *ck, *cv, *sk, *sv, *dx, *dy, *dq, *dattn, *dh, *logits, *first_logits, *splitk, *mask;
  char *tk_val, *tk_idx, *tk_stat;
  size_t splitk_bytes;
  DecodeState st;
};

struct Plan {   // workspace carve-up (all offsets 256-byte aligned)
  size_t total = 0;
  char *a1, *h1, *a2, *x, *y, *q, *k, *vt, *hbuf, *enc_out;
  int n_coh = 1;               // the windows of a call are decoded as 1 or 2 independent cohorts (see wseg_generate)
  DecPlan dec[2];
};

}  // namespace

struct wseg_model {
  wseg_model_config cfg;
  size_t es;                 // element size of the model dtype
  int kp1, vp, tp;           // conv1 K padded, vocab padded, encoder positions padded
  std::map<std::string, Slot> slots;
  const void *conv1_w, *conv1_b, *conv2_w, *conv2_b, *enc_pos, *enc_ln_g, *enc_ln_b;
  const void *dec_tok, *dec_pos, *dec_ln_g, *dec_ln_b;
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  hipEvent_t ev[4];
  bool ev_ok = false;
  int last_steps = 0;
  int last_W = 0, last_nb = 0, last_L = 0;
  bool timing_valid = false;
  int* poll = nullptr;       // pinned host mirror of DecodeState::flags (one int per decode position)
  int epoch = 0;
  // decode-step graph (hipGraph): captured once per (workspace, geometry, parameters), replayed per step
  hipGraphExec_t step_graph = nullptr;
  hipStream_t cap_stream = nullptr;   // capture happens on a private stream (the legacy NULL stream cannot capture)
  hipStream_t cap_stream2 = nullptr;  // second branch of the step graph (two cohorts)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<unsigned char> step_graph_key;
};

namespace {

void add_slot(wseg_model* m, const std::string& name, const void** field, size_t elems) {
  *field = nullptr;
  m->slots[name] = Slot{field, elems * m->es, false};
}

// Lay out the workspace for W windows, nb beams, capacity L positions.  base may be null (size query).
void make_plan(const wseg_model* m, int W, int nb, int L, char* base, Plan& p) {
  const wseg_model_config& c = m->cfg;
  const size_t es = m->es;
  const size_t d = c.d_model, H = c.n_heads, ffn = c.ffn;
  const size_t M1p = align_up((size_t)W * c.spec_cols, 256), Mp = align_up((size_t)W * c.enc_positions, 256);
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
  p.x = take(Mp * d * es);
  p.y = take(Mp * d * es);
  p.q = take((size_t)W * H * m->tp * 64 * es);
  p.k = take((size_t)W * H * m->tp * 64 * es);
  p.vt = take((size_t)W * H * m->tp * 64 * es);
  p.enc_out = take(Mp * d * es);
  // decoder: one cohort by default.  WSEG_TWO_COHORTS=1 splits the windows into two independent decode chains captured
  // as parallel branches of the step graph (windows never interact).  Measured on MI355X (large, 120 windows): the
  // branches do run concurrently, but 8.19 ms per step against 6.60 ms for one chain — the two chains compete for the
  // same CUs / LDS-DMA path instead of filling each other's gaps — so it stays an experiment.
  static const bool two_cohorts = getenv("WSEG_TWO_COHORTS") != nullptr;
  p.n_coh = (W >= 16 && two_cohorts) ? 2 : 1;
  const size_t Ld = c.dec_layers, Tk = c.enc_positions;
  const size_t maxn = 3 * d > ffn ? 3 * d : ffn;
  int w0 = 0;
  for (int ci = 0; ci < p.n_coh; ++ci) {
    DecPlan& q = p.dec[ci];
    q.W = p.n_coh == 1 ? W : (ci == 0 ? (W + 1) / 2 : W / 2);
    q.w0 = w0;
    w0 += q.W;
    const size_t Wc = q.W, R = Wc * nb, Rp = align_up(R, 256);
    q.ck = take(Ld * Wc * H * Tk * 64 * es);
    q.cv = take(Ld * Wc * H * Tk * 64 * es);
    q.sk = take(Ld * R * H * (size_t)L * 64 * es);
    q.sv = take(Ld * R * H * (size_t)L * 64 * es);
    q.dx = take(Rp * d * es);
    q.dy = take(Rp * d * es);
    q.dq = take(Rp * d * es);
    q.dattn = take(Rp * d * es);
    q.dh = take(Rp * ffn * es);
    q.logits = take(Rp * (size_t)m->vp * 4);
    q.first_logits = take(R * (size_t)m->vp * 4);
    q.splitk_bytes = (size_t)8 * Rp * maxn * 4;   // fp32 partials of the decoder-step GEMMs
    q.splitk = take(q.splitk_bytes);
    q.mask = take(align_up((size_t)c.vocab, 4));
    q.tk_val = take(R * 256 * 4);
    q.tk_idx = take(R * 256 * 4);
    q.tk_stat = take(R * 16 * 2 * 4);
    DecodeState& st = q.st;
    st.W = (int)Wc; st.nb = nb; st.L = L; st.V = c.vocab; st.ldv = m->vp;
    st.pos = (int*)take(256);
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
    st.active = (int*)take((size_t)L * 4);
    st.flags = (int*)take((size_t)L * 4);
    st.epoch = 0;
    st.sup_mask = (const unsigned char*)q.mask;
  }
  p.total = (size_t)(cur - base);
}

int check_geometry(const wseg_model_config& c) {
  if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model != c.n_heads * 64) { set_error("d_model %d must be n_heads %d * 64", c.d_model, c.n_heads); return WSEG_ERR_INVALID; }
  if (c.d_model % 128 || c.ffn % 128) { set_error("d_model/ffn must be multiples of 128"); return WSEG_ERR_INVALID; }
  if (c.spec_cols != 2 * c.enc_positions || c.enc_positions > 512) { set_error("spec_cols %d / enc_positions %d unsupported", c.spec_cols, c.enc_positions); return WSEG_ERR_INVALID; }
  if (c.n_mels <= 0 || c.n_mels > 96) { set_error("n_mels %d unsupported", c.n_mels); return WSEG_ERR_INVALID; }
  if (c.dec_positions <= 0 || c.dec_positions > 512) { set_error("dec_positions %d unsupported", c.dec_positions); return WSEG_ERR_INVALID; }
  if (c.dtype != WSEG_F32 && c.dtype != WSEG_BF16) { set_error("dtype %d unsupported", c.dtype); return WSEG_ERR_INVALID; }
  if (c.enc_layers <= 0 || c.dec_layers <= 0 || c.vocab <= 0) { set_error("bad layer/vocab counts"); return WSEG_ERR_INVALID; }
  return WSEG_OK;
}

#define WSEG_TRY(expr) do { int _s = (expr); if (_s != WSEG_OK) return _s; } while (0)

int gemm(const wseg_model* m, EpiKind epi, const void* A, int lda, const void* Wt, int ldw, int M, int N, int K,
         const EpiParams& ep, const DecPlan* p, hipStream_t s) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.ep = ep;
  if (p) { g.splitk_ws = (float*)p->splitk; g.splitk_ws_bytes = p->splitk_bytes; }
  return launch_gemm(m->cfg.dtype, epi, g, s);
}

int run_encoder(wseg_model* m, const float* feats, int W, Plan& p, void* enc_out, hipStream_t s) {
  const wseg_model_config& c = m->cfg;
  const int dt = c.dtype, d = c.d_model, H = c.n_heads, ffn = c.ffn, T = c.enc_positions, Tp = m->tp;
  const int M1 = W * c.spec_cols, M = W * T;
  const size_t qkv_bytes = (size_t)W * H * Tp * 64 * m->es;
  // pad rows of Q/K and pad columns of V^T must be finite (they are multiplied by exact zeros)
  WSEG_HIP_CHECK(hipMemsetAsync(p.q, 0, qkv_bytes, s));
  WSEG_HIP_CHECK(hipMemsetAsync(p.k, 0, qkv_bytes, s));
  WSEG_HIP_CHECK(hipMemsetAsync(p.vt, 0, qkv_bytes, s));
  WSEG_TRY(launch_im2col_conv1(dt, feats, p.a1, W, c.n_mels, c.spec_cols, m->kp1, s));
  EpiParams e;
  e.bias = m->conv1_b; e.out = p.h1; e.ldc = d;
  WSEG_TRY(gemm(m, EPI_GELU, p.a1, m->kp1, m->conv1_w, m->kp1, M1, d, m->kp1, e, nullptr, s));
  WSEG_TRY(launch_im2col_conv2(dt, p.h1, p.a2, W, c.spec_cols, d, s));
  e = EpiParams();
  e.bias = m->conv2_b; e.out = p.x; e.ldc = d; e.pos = m->enc_pos; e.pos_rows = T;
  WSEG_TRY(gemm(m, EPI_GELU_POS, p.a2, 3 * d, m->conv2_w, 3 * d, M, d, 3 * d, e, nullptr, s));
  for (int l = 0; l < c.enc_layers; ++l) {
    const EncLayer& L = m->enc[l];
    WSEG_TRY(launch_layernorm(dt, p.x, L.ln1_g, L.ln1_b, p.y, M, d, s));
    e = EpiParams();
    e.bias = L.qkv_b; e.q = p.q; e.k = p.k; e.v = p.vt; e.d_model = d; e.t_len = T; e.t_pad = Tp; e.n_heads = H; e.scale = 0.125f;
    WSEG_TRY(gemm(m, EPI_QKV_ENC, p.y, d, L.qkv_w, d, M, 3 * d, d, e, nullptr, s));
    WSEG_TRY(launch_enc_attention(dt, p.q, p.k, p.vt, p.y, W, H, T, Tp, d, s));
    e = EpiParams();
    e.bias = L.o_b; e.out = p.x; e.resid = p.x; e.ldc = d;
    WSEG_TRY(gemm(m, EPI_RESID, p.y, d, L.o_w, d, M, d, d, e, nullptr, s));
    WSEG_TRY(launch_layernorm(dt, p.x, L.ln2_g, L.ln2_b, p.y, M, d, s));
    e = EpiParams();
    e.bias = L.fc1_b; e.out = p.hbuf; e.ldc = ffn;
    WSEG_TRY(gemm(m, EPI_GELU, p.y, d, L.fc1_w, d, M, ffn, d, e, nullptr, s));
    e = EpiParams();
    e.bias = L.fc2_b; e.out = p.x; e.resid = p.x; e.ldc = d;
    WSEG_TRY(gemm(m, EPI_RESID, p.hbuf, ffn, L.fc2_w, ffn, M, d, ffn, e, nullptr, s));
  }
  WSEG_TRY(launch_layernorm(dt, p.x, m->enc_ln_g, m->enc_ln_b, enc_out, M, d, s));
  return WSEG_OK;
}

// One decoder step for all R rows at position *st.pos.  want_logits: run final LN + LM head.
int run_decoder_step(wseg_model* m, DecPlan& p, bool want_logits, hipStream_t s) {
  const wseg_model_config& c = m->cfg;
  const int dt = c.dtype, d = c.d_model, H = c.n_heads, ffn = c.ffn, Tk = c.enc_positions;
  const DecodeState& st = p.st;
  const int R = st.W * st.nb;
  const size_t es = m->es;
  const size_t self_stride = (size_t)R * H * st.L * 64 * es;
  const size_t cross_stride = (size_t)st.W * H * Tk * 64 * es;
  WSEG_TRY(launch_embed(dt, st, m->dec_tok, m->dec_pos, p.dx, d, s));
  EpiParams e;
  auto gemm_resid_ln = [&](const void* A, int K, const void* Wt, const void* bias, const void* g_, const void* b_) -> int {
    GemmArgs g;
    g.A = A; g.lda = K; g.W = Wt; g.ldw = K; g.M = R; g.N = d; g.K = K;
    g.ep.bias = bias; g.ep.out = p.dx; g.ep.resid = p.dx; g.ep.ldc = d;
    g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
    return launch_gemm_resid_ln(dt, g, g_, b_, p.dy, s);
  };
  // y = LN1(x) of layer 0; every later LayerNorm is fused into the reduction of the GEMM that precedes it
  WSEG_TRY(launch_layernorm(dt, p.dx, m->dec[0].ln1_g, m->dec[0].ln1_b, p.dy, R, d, s));
  for (int l = 0; l < c.dec_layers; ++l) {
    const DecLayer& L = m->dec[l];
    {   // q|k|v projection: partial sums only when possible; the attention kernel finishes the reduction
      GemmArgs g;
      g.A = p.dy; g.lda = d; g.W = L.qkv_w; g.ldw = d; g.M = R; g.N = 3 * d; g.K = d;
      g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
      PartialInfo pi; bool ok = false;
      WSEG_TRY(launch_gemm_partial(dt, g, &pi, &ok, s));
      if (ok) {
        WSEG_TRY(launch_dec_self_attn(dt, st, nullptr, p.sk + l * self_stride, p.sv + l * self_stride, p.dattn, H, d, &pi, L.qkv_b, 0.125f, s));
      } else {
        e = EpiParams();
        e.bias = L.qkv_b; e.q = p.dq; e.k = p.sk + l * self_stride; e.v = p.sv + l * self_stride;
        e.d_model = d; e.n_heads = H; e.t_pad = st.L; e.pos_ptr = st.pos; e.scale = 0.125f;
        WSEG_TRY(gemm(m, EPI_QKV_DEC, p.dy, d, L.qkv_w, d, R, 3 * d, d, e, &p, s));
        WSEG_TRY(launch_dec_self_attn(dt, st, p.dq, p.sk + l * self_stride, p.sv + l * self_stride, p.dattn, H, d, nullptr, nullptr, 0.125f, s));
      }
    }
    WSEG_TRY(gemm_resid_ln(p.dattn, d, L.o_w, L.o_b, L.ln2_g, L.ln2_b));                    // x += attn Wo ; y = LN2(x)
    {   // cross-attention query: same scheme
      GemmArgs g;
      g.A = p.dy; g.lda = d; g.W = L.cq_w; g.ldw = d; g.M = R; g.N = d; g.K = d;
      g.splitk_ws = (float*)p.splitk; g.splitk_ws_bytes = p.splitk_bytes;
      PartialInfo pi; bool ok = false;
      WSEG_TRY(launch_gemm_partial(dt, g, &pi, &ok, s));
      if (ok) {
        WSEG_TRY(launch_dec_cross_attn(dt, st, nullptr, p.ck + l * cross_stride, p.cv + l * cross_stride, p.dattn, H, Tk, d, &pi, L.cq_b, 0.125f, s));
      } else {
        e = EpiParams();
        e.bias = L.cq_b; e.out = p.dq; e.ldc = d; e.scale = 0.125f;
        WSEG_TRY(gemm(m, EPI_SCALE, p.dy, d, L.cq_w, d, R, d, d, e, &p, s));
        WSEG_TRY(launch_dec_cross_attn(dt, st, p.dq, p.ck + l * cross_stride, p.cv + l * cross_stride, p.dattn, H, Tk, d, nullptr, nullptr, 0.125f, s));
      }
    }
    WSEG_TRY(gemm_resid_ln(p.dattn, d, L.co_w, L.co_b, L.ln3_g, L.ln3_b));                  // x += cross Wo ; y = LN3(x)
    e = EpiParams();
    e.bias = L.fc1_b; e.out = p.dh; e.ldc = ffn;
    WSEG_TRY(gemm(m, EPI_GELU, p.dy, d, L.fc1_w, d, R, ffn, d, e, &p, s));
    const bool last = l + 1 == c.dec_layers;                                                 // x += fc2 ; y = next LN1 / final LN
    WSEG_TRY(gemm_resid_ln(p.dh, ffn, L.fc2_w, L.fc2_b, last ? m->dec_ln_g : m->dec[l + 1].ln1_g,
                           last ? m->dec_ln_b : m->dec[l + 1].ln1_b));
  }
  if (want_logits) {
    e = EpiParams();
    e.out_f32 = (float*)p.logits; e.ldc = m->vp;
    WSEG_TRY(gemm(m, EPI_F32, p.dy, d, m->dec_tok, d, R, m->vp, d, e, nullptr, s));
  }
  return WSEG_OK;
}

}  // namespace

extern "C" int wseg_model_create(const wseg_model_config* cfg, wseg_model** out) {
  if (!cfg || !out) { set_error("wseg_model_create: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(check_geometry(*cfg));
  wseg_model* m = new wseg_model();
  m->cfg = *cfg;
  m->es = cfg->dtype == WSEG_BF16 ? 2 : 4;
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

extern "C" void wseg_model_destroy(wseg_model* m) {
  if (!m) return;
  if (m->ev_ok) for (int i = 0; i < 4; ++i) (void)hipEventDestroy(m->ev[i]);
  if (m->step_graph) (void)hipGraphExecDestroy(m->step_graph);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  if (m->cap_stream2) (void)hipStreamDestroy(m->cap_stream2);
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  if (m->ev_join) (void)hipEventDestroy(m->ev_join);
  if (m->poll) (void)hipHostFree(m->poll);
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
  return WSEG_OK;
}

extern "C" int wseg_model_ready(const wseg_model* m) {
  if (!m) { set_error("wseg_model_ready: null model"); return WSEG_ERR_INVALID; }
  for (const auto& kv : m->slots)
    if (!kv.second.set) { set_error("tensor '%s' has not been attached", kv.first.c_str()); return WSEG_ERR_STATE; }
  return WSEG_OK;
}

extern "C" size_t wseg_workspace_bytes(const wseg_model* m, int32_t max_windows, int32_t num_beams, int32_t max_length) {
  if (!m || max_windows <= 0 || num_beams <= 0 || num_beams > MAX_BEAMS || max_length <= 0) return 0;
  Plan p;
  make_plan(m, max_windows, num_beams, max_length, nullptr, p);
  return p.total + 256;
}

static char* aligned_base(void* ws) { return (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255); }

extern "C" int wseg_encode(wseg_model* m, const float* feats, int32_t n_windows, void* workspace, size_t workspace_bytes,
                           void* enc_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !feats || !workspace || !enc_out) { set_error("wseg_encode: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(wseg_model_ready(m));
  if (n_windows <= 0) return WSEG_OK;
  Plan p;
  make_plan(m, n_windows, 1, 8, aligned_base(workspace), p);
  if (p.total + 256 > workspace_bytes) { set_error("workspace too small: need %zu, have %zu", p.total + 256, workspace_bytes); return WSEG_ERR_STATE; }
  WSEG_TRY(run_encoder(m, feats, n_windows, p, p.enc_out, s));
  const size_t bytes = (size_t)n_windows * m->cfg.enc_positions * m->cfg.d_model * m->es;
  WSEG_HIP_CHECK(hipMemcpyAsync(enc_out, p.enc_out, bytes, hipMemcpyDeviceToDevice, s));
  return WSEG_OK;
}

extern "C" int wseg_generate(wseg_model* m, const float* feats, int32_t n_windows, const wseg_generate_params* gp,
                             void* workspace, size_t workspace_bytes, int32_t* out_tokens, int32_t* out_lengths,
                             void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !feats || !gp || !workspace || !out_tokens || !out_lengths) { set_error("wseg_generate: null argument"); return WSEG_ERR_INVALID; }
  WSEG_TRY(wseg_model_ready(m));
  if (n_windows <= 0) return WSEG_OK;
  const wseg_model_config& c = m->cfg;
  const int nb = gp->num_beams, P = gp->prompt_len, L = gp->max_length;
  if (nb < 1 || nb > MAX_BEAMS) { set_error("num_beams %d unsupported (1..%d)", nb, MAX_BEAMS); return WSEG_ERR_INVALID; }
  if (P < 1 || P > 8 || L <= P || L > c.dec_positions) { set_error("prompt_len %d / max_length %d unsupported", P, L); return WSEG_ERR_INVALID; }
  if (gp->n_suppress < 0 || gp->n_begin_suppress < 0 || (gp->n_suppress && !gp->suppress_tokens) || (gp->n_begin_suppress && !gp->begin_suppress_tokens)) {
    set_error("bad suppress-token lists"); return WSEG_ERR_INVALID;
  }
  Plan p;
  make_plan(m, n_windows, nb, L, aligned_base(workspace), p);
  if (p.total + 256 > workspace_bytes) { set_error("workspace too small: need %zu, have %zu", p.total + 256, workspace_bytes); return WSEG_ERR_STATE; }
  if (!m->ev_ok) {
    for (int i = 0; i < 4; ++i) WSEG_HIP_CHECK(hipEventCreate(&m->ev[i]));
    m->ev_ok = true;
  }
  if (!m->poll) WSEG_HIP_CHECK(hipHostMalloc((void**)&m->poll, 2 * 512 * sizeof(int), hipHostMallocDefault));
  m->epoch = (m->epoch % 100000000) + 1;
  for (int ci = 0; ci < p.n_coh; ++ci) {
    DecodeState& st = p.dec[ci].st;
    st.P = P; st.eos = gp->eos_token_id; st.pad = gp->pad_token_id; st.max_length = L; st.length_penalty = gp->length_penalty;
    for (int i = 0; i < 8; ++i) st.prompt[i] = i < P ? gp->prompt[i] : 0;
    st.epoch = m->epoch;
  }

  m->timing_valid = false;
  WSEG_HIP_CHECK(hipEventRecord(m->ev[0], s));
  WSEG_TRY(run_encoder(m, feats, n_windows, p, p.enc_out, s));
  WSEG_HIP_CHECK(hipEventRecord(m->ev[1], s));
  // cross-attention K/V of every decoder layer, once per window (shared by its beams), per cohort
  for (int ci = 0; ci < p.n_coh; ++ci) {
    DecPlan& q = p.dec[ci];
    const int d = c.d_model, H = c.n_heads, Tk = c.enc_positions, M = q.W * Tk;
    const size_t cross_stride = (size_t)q.W * H * Tk * 64 * m->es;
    const char* enc_rows = p.enc_out + (size_t)q.w0 * Tk * d * m->es;
    for (int l = 0; l < c.dec_layers; ++l) {
      EpiParams e;
      e.bias = m->dec[l].ckv_b; e.k = q.ck + l * cross_stride; e.v = q.cv + l * cross_stride;
      e.d_model = d; e.t_len = Tk; e.n_heads = H;
      WSEG_TRY(gemm(m, EPI_KV_CROSS, enc_rows, d, m->dec[l].ckv_w, d, M, 2 * d, d, e, nullptr, s));
    }
  }
  WSEG_HIP_CHECK(hipEventRecord(m->ev[2], s));
  for (int ci = 0; ci < p.n_coh; ++ci) {
    WSEG_TRY(launch_build_suppress_mask((unsigned char*)p.dec[ci].mask, c.vocab, gp->suppress_tokens, gp->n_suppress,
                                        gp->begin_suppress_tokens, gp->n_begin_suppress, s));
    WSEG_TRY(launch_decode_init(p.dec[ci].st, s));
  }

  // One generated-token step of one cohort: decoder layers, LM head, candidates, bookkeeping, advance, verdict mirror.
  auto enqueue_gen_step = [&](int ci, bool snapshot_logits, hipStream_t qs) -> int {
    DecPlan& q = p.dec[ci];
    WSEG_TRY(run_decoder_step(m, q, true, qs));
    if (snapshot_logits)
      WSEG_HIP_CHECK(hipMemcpyAsync(q.first_logits, q.logits, (size_t)q.W * nb * m->vp * 4, hipMemcpyDeviceToDevice, qs));
    WSEG_TRY(launch_row_topk(q.st, (const float*)q.logits, (float*)q.tk_val, (int*)q.tk_idx, (float*)q.tk_stat, qs));
    if (nb == 1) WSEG_TRY(launch_greedy_step(q.st, qs));
    else WSEG_TRY(launch_beam_step(q.st, qs));
    WSEG_TRY(launch_advance(q.st, qs));
    WSEG_HIP_CHECK(hipMemcpyAsync(m->poll + ci * 512, q.st.flags, (size_t)L * sizeof(int), hipMemcpyDeviceToHost, qs));
    return WSEG_OK;
  };
  // The step reads every step-dependent value (position, tokens, ancestry, epoch) from device memory, so ONE captured
  // graph serves all steps and later calls: replay costs ~1.6 us per kernel instead of ~5 us per eager launch.  With two
  // cohorts the graph has two parallel branches (fork / join through events on two capture streams).
  static const bool use_graph = getenv("WSEG_NO_GRAPH") == nullptr;
  std::vector<unsigned char> key;
  {
    auto put = [&](const void* ptr, size_t n) { const unsigned char* b = (const unsigned char*)ptr; key.insert(key.end(), b, b + n); };
    void* base = aligned_base(workspace);
    const DecodeState& st = p.dec[0].st;
    put(&base, sizeof(base)); put(&n_windows, 4); put(&nb, 4); put(&L, 4); put(&p.n_coh, 4);
    put(&st.P, 4); put(&st.eos, 4); put(&st.pad, 4); put(&st.length_penalty, 4); put(st.prompt, sizeof(st.prompt));
  }
  int steps = 0;
  bool stop = false;
  auto check_stop = [&](int upto) {
    int done = 0;
    for (int ci = 0; ci < p.n_coh; ++ci) {
      const volatile int* pl = (const volatile int*)(m->poll + ci * 512);
      for (int t = P - 1; t < upto; ++t)
        if (pl[t] == m->epoch * 4 + 2) { ++done; break; }
    }
    if (done == p.n_coh) stop = true;
  };
  for (int t = 0; t < L - 1 && !stop; ++t) {
    if (t < P - 1) {
      for (int ci = 0; ci < p.n_coh; ++ci) {
        WSEG_TRY(run_decoder_step(m, p.dec[ci], false, s));
        WSEG_TRY(launch_prompt_feed(p.dec[ci].st, s));
        WSEG_TRY(launch_advance(p.dec[ci].st, s));
      }
    } else if (t == P - 1 || !use_graph) {
      for (int ci = 0; ci < p.n_coh; ++ci) WSEG_TRY(enqueue_gen_step(ci, t == P - 1, s));
    } else {
      if (!m->step_graph || m->step_graph_key != key) {
        if (m->step_graph) { (void)hipGraphExecDestroy(m->step_graph); m->step_graph = nullptr; }
        hipGraph_t graph = nullptr;
        if (!m->cap_stream) WSEG_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        if (!m->cap_stream2) {
          WSEG_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream2, hipStreamNonBlocking));
          WSEG_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
          WSEG_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
        }
        WSEG_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = WSEG_OK;
        if (p.n_coh == 2) {
          hipError_t e1 = hipEventRecord(m->ev_fork, m->cap_stream);
          hipError_t e2 = hipStreamWaitEvent(m->cap_stream2, m->ev_fork, 0);
          if (e1 != hipSuccess || e2 != hipSuccess) rc = WSEG_ERR_HIP;
          if (rc == WSEG_OK) rc = enqueue_gen_step(1, false, m->cap_stream2);
          if (rc == WSEG_OK) rc = enqueue_gen_step(0, false, m->cap_stream);
          e1 = hipEventRecord(m->ev_join, m->cap_stream2);
          e2 = hipStreamWaitEvent(m->cap_stream, m->ev_join, 0);
          if (rc == WSEG_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = WSEG_ERR_HIP;
        } else {
          rc = enqueue_gen_step(0, false, m->cap_stream);
        }
        const hipError_t ec = hipStreamEndCapture(m->cap_stream, &graph);
        if (rc != WSEG_OK) { if (graph) (void)hipGraphDestroy(graph); if (rc == WSEG_ERR_HIP) set_error("graph capture fork/join failed"); return rc; }
        if (ec != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ec)); return WSEG_ERR_HIP; }
        WSEG_HIP_CHECK(hipGraphInstantiate(&m->step_graph, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        m->step_graph_key = key;
      }
      WSEG_HIP_CHECK(hipGraphLaunch(m->step_graph, s));
    }
    ++steps;
    // lagged, non-blocking poll of the pinned verdict mirror: stop enqueueing once a finished step of EVERY cohort reported
    // that no window can still improve (steps already enqueued are harmless: finished beams are frozen)
    check_stop(t);
  }
  for (int ci = 0; ci < p.n_coh; ++ci)
    WSEG_TRY(launch_finalize(p.dec[ci].st, out_tokens + (size_t)p.dec[ci].w0 * L, out_lengths + p.dec[ci].w0, s));
  WSEG_HIP_CHECK(hipEventRecord(m->ev[3], s));
  m->last_steps = steps;
  m->last_W = n_windows; m->last_nb = nb; m->last_L = L;
  m->timing_valid = true;
  return WSEG_OK;
}

extern "C" int wseg_debug_first_logits(wseg_model* m, void* workspace, float* out, int32_t n_rows, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!m || !workspace || !out || n_rows <= 0) { set_error("wseg_debug_first_logits: bad argument"); return WSEG_ERR_INVALID; }
  if (m->last_W <= 0 || n_rows > m->last_W * m->last_nb) { set_error("no matching wseg_generate call"); return WSEG_ERR_STATE; }
  Plan p;
  make_plan(m, m->last_W, m->last_nb, m->last_L, aligned_base(workspace), p);
  for (int ci = 0; ci < p.n_coh; ++ci) {
    const DecPlan& q = p.dec[ci];
    const int r0 = q.w0 * m->last_nb;
    int rows = q.W * m->last_nb;
    if (r0 >= n_rows) break;
    if (r0 + rows > n_rows) rows = n_rows - r0;
    WSEG_HIP_CHECK(hipMemcpy2DAsync(out + (size_t)r0 * m->cfg.vocab, (size_t)m->cfg.vocab * 4, q.first_logits, (size_t)m->vp * 4,
                                    (size_t)m->cfg.vocab * 4, (size_t)rows, hipMemcpyDeviceToDevice, s));
  }
  return WSEG_OK;
}

extern "C" int wseg_last_timing(const wseg_model* m, float out[4]) {
  if (!m || !out) { set_error("wseg_last_timing: null argument"); return WSEG_ERR_INVALID; }
  if (!m->timing_valid) { set_error("no completed wseg_generate call to time"); return WSEG_ERR_STATE; }
  for (int i = 0; i < 3; ++i) {
    WSEG_HIP_CHECK(hipEventSynchronize(m->ev[i + 1]));
    WSEG_HIP_CHECK(hipEventElapsedTime(&out[i], m->ev[i], m->ev[i + 1]));
  }
  out[3] = (float)m->last_steps;
  return WSEG_OK;
}
