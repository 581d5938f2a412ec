// CTC prefix beam search: the per-frame prefix bookkeeping (extend / merge / sort / prune) as native host code.
//
// Reference: CTCDecoder._beam_search and _merge_ctc_paths, asr/modeling/decoders/ctc.py:262-344, 372-397 -- python floats
// (IEEE doubles), a dict keyed by the label tuple, sorted(..., reverse=True).  The arithmetic here is the same sequence of double
// operations (log-sum-exp of two terms with the larger one factored out, sums in the reference's order), the merge keeps the FIRST
// path's LM / length scores and folds the path probabilities, the sort is stable: results are bit-identical to the Python
// restatement in emoasr_amd/modeling/ctc_beam_search.py (EMOASR_CTC_BEAM_NATIVE=0 selects that one; tests compare the two).
//
// What stays outside: the acoustic side (projection, log-soft-max, per-frame top-k: device kernels, one copy per utterance) and the
// Transformer LM (device; one call per NEW prefix, rows cached on the device, only the k candidate columns of a frame come to the
// host).  A frame's bookkeeping is ~110 candidates: 20 us here against ~1 ms of interpreter time.
#include <math.h>
#include <algorithm>
#include <map>
#include <vector>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

constexpr double NEG = -1e10;   // LOG_0 of decoders/ctc.py:23

inline double lse2(double a, double b) {
  const double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

struct Prefix {
  std::vector<int> toks;
  double p_b, p_nb, asr, lm, len_bonus;
  int n_plain;      // tokens that are not <eos>
  int parent, tok;  // how the last step formed it: index of the parent in the previous live set, appended label (-1: unchanged)
  double total() const { return asr + lm + len_bonus; }
};

struct Beam {
  int width, blank, eos;
  double len_weight, lm_weight;
  std::vector<Prefix> live;
};

}  // namespace

struct emoasr_ctc_beam { Beam b; };

extern "C" emoasr_ctc_beam_t* emoasr_ctc_beam_new(int beam_width, int blank, int eos, double len_weight, double lm_weight) {
  if (beam_width < 1) return nullptr;
  emoasr_ctc_beam_t* h = new emoasr_ctc_beam_t();
  h->b.width = beam_width; h->b.blank = blank; h->b.eos = eos;
  h->b.len_weight = len_weight; h->b.lm_weight = lm_weight;
  h->b.live.push_back(Prefix{{eos}, 0.0, NEG, 0.0, 0.0, 0.0, 0, -1, -1});   // hypotheses start with <eos> (the LM's BOS)
  return h;
}

extern "C" void emoasr_ctc_beam_free(emoasr_ctc_beam_t* h) { delete h; }

extern "C" int emoasr_ctc_beam_size(const emoasr_ctc_beam_t* h) { return h ? (int)h->b.live.size() : 0; }

// One frame.  row: the frame's log-probabilities (f32 [V], read as doubles); top: the frame's k best labels, best first (the blank
// among them is skipped); lm_lp: [live, k] LM log-probabilities of those labels after each live prefix (NULL: no LM).
// -> the new number of live prefixes; parent / tok [beam_width]: how each new prefix was formed (tok = -1: the parent itself).
extern "C" int emoasr_ctc_beam_step(emoasr_ctc_beam_t* h, const float* row, const int* top, int k, const double* lm_lp,
                                    int* parent, int* tok) {
  if (!(h && row && top && k >= 0)) { emo_set_error("ctc_beam_step: missing arguments"); return -1; }
  Beam& B = h->b;
  const double lp_blank = (double)row[B.blank];
  const bool use_lm = lm_lp != nullptr && B.lm_weight > 0;
  std::vector<Prefix> order;                       // first-seen order, as the reference's dict iterates
  order.reserve(B.live.size() * (size_t)(k + 1));
  std::map<std::vector<int>, int> table;           // label sequence -> index in `order`
  auto put = [&](Prefix&& q) {
    auto it = table.find(q.toks);
    if (it == table.end()) {
      table.emplace(q.toks, (int)order.size());
      order.push_back(std::move(q));
    } else {   // the same label sequence reached twice: fold the path probabilities only (ctc.py:388-393)
      Prefix& old = order[it->second];
      old.p_b = lse2(old.p_b, q.p_b);
      old.p_nb = lse2(old.p_nb, q.p_nb);
      old.asr = lse2(old.asr, q.asr);
    }
  };
  for (int i = 0; i < (int)B.live.size(); ++i) {
    const Prefix& p = B.live[i];
    const bool has_last = p.toks.size() > 1;
    const int last = has_last ? p.toks.back() : -1;
    // stay on the same prefix: blank, or a repeat of its last label
    const double stay_b = lse2(p.p_b + lp_blank, p.p_nb + lp_blank);
    const double stay_nb = has_last ? p.p_nb + (double)row[last] : NEG;
    put(Prefix{p.toks, stay_b, stay_nb, lse2(stay_b, stay_nb), p.lm, p.len_bonus, p.n_plain, i, -1});
    // extend by each of the frame's top-k labels; the LM score of the j-th candidate also carries those of the candidates
    // tried before it (ctc.py:309-310), the length bonus counts the PARENT's labels (ctc.py:308)
    double lm_run = p.lm;
    const double bonus = B.len_weight * (double)(p.n_plain + 1);
    for (int j = 0; j < k; ++j) {
      const int v = top[j];
      if (v == B.blank) continue;
      const double lp_v = (double)row[v];
      const double ext_nb = (has_last && v == last) ? p.p_b + lp_v : lse2(p.p_b + lp_v, p.p_nb + lp_v);
      if (use_lm) lm_run += B.lm_weight * lm_lp[(size_t)i * k + j];
      std::vector<int> t2 = p.toks;
      t2.push_back(v);
      put(Prefix{std::move(t2), NEG, ext_nb, lse2(NEG, ext_nb), lm_run, bonus, p.n_plain + (v == B.eos ? 0 : 1), i, v});
    }
  }
  std::stable_sort(order.begin(), order.end(), [](const Prefix& a, const Prefix& b) { return a.total() > b.total(); });
  if ((int)order.size() > B.width) order.resize(B.width);
  B.live.swap(order);
  for (int i = 0; i < (int)B.live.size(); ++i) {
    if (parent) parent[i] = B.live[i].parent;
    if (tok) tok[i] = B.live[i].tok;
  }
  return (int)B.live.size();
}

// labels of live prefix i (<= cap written) -> its length; scores (asr + lm + length bonus) of all live prefixes
extern "C" int emoasr_ctc_beam_prefix(const emoasr_ctc_beam_t* h, int i, int* toks, int cap) {
  if (!h || i < 0 || i >= (int)h->b.live.size()) return -1;
  const std::vector<int>& t = h->b.live[i].toks;
  for (int j = 0; j < (int)t.size() && j < cap; ++j) toks[j] = t[j];
  return (int)t.size();
}

extern "C" int emoasr_ctc_beam_scores(const emoasr_ctc_beam_t* h, double* total) {
  if (!h || !total) return 0;
  for (size_t i = 0; i < h->b.live.size(); ++i) total[i] = h->b.live[i].total();
  return (int)h->b.live.size();
}
