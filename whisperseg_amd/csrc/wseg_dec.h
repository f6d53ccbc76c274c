// Decoder-side kernels and the device-resident decoding state (internal to libwseg).
#pragma once
#include "wseg_kernels.h"

namespace wseg {

constexpr int MAX_BEAMS = 8;
constexpr int MAX_CAND = 2 * MAX_BEAMS;

// Everything the decode loop needs lives on the device; the host only enqueues kernels and reads a small
// per-step status mirror.  W window SLOTS, nb beams, R = W * nb rows, L = max_length (cache capacity).
// Slots are independent: every slot has its own decode position, so a finished window's slot can be
// re-used for the next queued window while its neighbours keep decoding (wseg_generate's scheduler).
struct DecodeState {
  int W, nb, L, V, ldv;            // ldv: leading dimension of the logits buffer (vocab padded)
  int P;                           // prompt length
  int eos, pad, max_length;
  float length_penalty;
  int prompt[8];
  int* pos;                        // [W]   position of the token each slot feeds this step
  int* done;                       // [W]   1 = slot idle (free, or finished and waiting to be retired): every kernel skips it
  int* win;                        // [W]   index of the window decoded in the slot (row of the output arrays)
  int* wmax;                       // [W]   total-length cap of the slot's window (<= max_length)
  const int* win_max_length;       // [n_windows] per-window caps (device) or null
  int top_k;                       // > 1 (greedy path only): sample among the top_k processed logits
  float top_p;                     // nucleus mass for sampling
  const unsigned long long* seed;  // device: sampling seed of the current call (read by greedy_step_kernel; not part of the step graph)
  int* tokens_in;                  // [R]   token fed at this step
  int* run_seq;                    // [W][nb][L]
  int* fin_seq;                    // [W][nb][L]
  float* run_score;                // [W][nb]
  float* fin_score;                // [W][nb]
  int* fin_flag;                   // [W][nb]
  int* fin_len;                    // [W][nb]  generated length of each finished slot
  int* unsat;                      // [W]      is_early_stop_heuristic_unsatisfied (beam) / unfinished (greedy)
  const int* kv_pt;                // [W][npg] pool unit holding positions [KV_PAGE k, KV_PAGE (k + 1)) of the slot's self K / V
  int npg;                         // page-table entries per slot = ceil(L / KV_PAGE)
  unsigned char* anc;              // [W][nb][L] cache slot (beam index) that holds position p of this row's history
  float* cand_val;                 // [R][Kc]
  int* cand_tok;                   // [R][Kc]
  const unsigned char* sup_mask;   // [V] bit0: always suppressed, bit1: suppressed at the first generated position
};

// all slots idle (start of a wseg_generate call)
int launch_decode_reset(const DecodeState& st, hipStream_t s);
// slots[i] starts decoding window wins[i] at position pf_np (device arrays of n entries); pf_np > 0: the prompt positions before it
// were run by the prompt pass (launch_prompt_*), every beam's ancestry of them points at beam 0's cache rows
// pos0 <= pf_np: the position the slot's state is left at (pf_np == P, pos0 == P - 1: the prompt pass also runs position P - 1 and the
// admission finishes the first generated step itself — launch_row_topk / launch_beam_step on the admitted list)
int launch_decode_admit(const DecodeState& st, const int* slots, const int* wins, int n, int pf_np, int pos0, hipStream_t s);
// Prompt pass (split-precision modes): the first np <= 4 forced prompt positions of n admitted windows as ONE pass of n * np rows
// (row i * np + pp) instead of np decode steps of every slot — the cross-attention K / V of a window are streamed once for them.
// x[row][:] = tok_emb[prompt[pp]][:] + pos_emb[pp][:]
int launch_prompt_embed(int dtype, const DecodeState& st, int rows, int np, const void* tok_emb, const void* pos_emb, void* x, int d, hipStream_t s);
// causal self-attention among the np positions of a window; K / V -> beam 0's rows of the first page of slots[i].  q | k | v: split-K
// partials (qkv_part) or fp32 rows [rows][3 d] WITHOUT bias (qkv); out: the o-proj GEMM's operand rows.
int launch_prompt_self_attn(int dtype, const DecodeState& st, const float* qkv, const PartialInfo* qkv_part, const void* qkv_bias, void* kc, void* vc,
                            const int* slots, int n, int np, void* out, int H, int d, float scale, hipStream_t s);
// page-table updates: kv_pt[pairs[2 i]] = pairs[2 i + 1] for i < n (device array of 2 n ints, written by the scheduler)
int launch_kv_assign(int* kv_pt, const int* pairs, int n, hipStream_t s);
// preemption: slots[i] stops decoding and produces no output (its window is re-queued by the scheduler)
int launch_decode_abort(const DecodeState& st, const int* slots, int n, hipStream_t s);
int launch_build_suppress_mask(unsigned char* mask, int V, const int* sup, int n_sup, const int* bsup, int n_bsup, hipStream_t s);
// x[r][:] = tok_emb[tokens_in[r]][:] + pos_emb[pos[slot of r]][:]   (x: fp32 residual stream)
int launch_embed(int dtype, const DecodeState& st, const void* tok_emb, const void* pos_emb, void* x, int d, hipStream_t s);
// decoder self-attention over the paged KV cache (pool of one layer: [units][nb][H][KV_PAGE][64], st.kv_pt) with per-position ancestry
// qkv_part != nullptr: q/k/v of this step arrive as split-K partials [z][m_pad][3d] (+ qkv_bias); the kernel finishes the
// reduction, appends k/v to the cache and uses them (saves the separate reduction launch).
int launch_dec_self_attn(int dtype, const DecodeState& st, const void* q, void* kc, void* vc, void* out,
                         int H, int d, const PartialInfo* qkv_part, const void* qkv_bias, float scale, hipStream_t s);
// cross-attention: the nb beams of a window share K/V [W][H][Tk][64]
// q_part != nullptr: the query arrives as split-K partials [z][m_pad][d] (+ q_bias, scaled by `scale`)
// kv_slot != nullptr (prompt pass, block-floating-point K / V only): "window" w of st is admitted window w with st.nb = np query
// rows, its K / V are those of slot kv_slot[w]
int launch_dec_cross_attn(int dtype, const DecodeState& st, const void* q, const void* ck, const void* cv, void* out,
                          int H, int Tk, int d, const PartialInfo* q_part, const void* q_bias, float scale, hipStream_t s,
                          const int* kv_slot = nullptr);
// WSEG_F16M6: M6 rows (24-bit K / V kernel, <= 4 beams) or hi | lo rows that the caller converts (5..8 beams)?
bool dec_cross_attn_writes_mx(int dtype, int nb);
// log-softmax + suppress + running score -> top-2nb per row (beam) / argmax of the processed logits (greedy)
// scratch: part_val/part_idx [R][16][16], part_stat [R][16][2]
// list != nullptr: only the n_list slots named there (device array), logits row i = the ONE row of window list[i] (first generated step of
// newly admitted windows: their beams are copies)
int launch_row_topk(const DecodeState& st, const float* logits, float* part_val, int* part_idx, float* part_stat, hipStream_t s,
                    const int* list = nullptr, int n_list = 0);
int launch_beam_step(const DecodeState& st, hipStream_t s, const int* list = nullptr, int n_list = 0);
int launch_greedy_step(const DecodeState& st, hipStream_t s, const int* list = nullptr, int n_list = 0);
// retire: best sequence of slots[i] -> out_tokens[win[slots[i]]][:], out_lengths[win[slots[i]]]
int launch_finalize(const DecodeState& st, const int* slots, int n, int* out_tokens, int* out_lengths, hipStream_t s);

}  // namespace wseg
