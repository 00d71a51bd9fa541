// hinted_long.hpp -- pass 0 for reads of 161 to 256 bases (2 x 250): EIGHT k-mers, rows of 128 bytes.
//
// The same decisions as hinted.hpp's hinted_on_path -- the same proofs per k-mer (hint_kmer, the filter probes), the same run
// selection, walks inside a reference node or over one SNP site at either end, twin paths, two runs of one length, the same record --
// written over arrays of NK values and chains of selects instead of five named registers.  (The five-k-mer kernel keeps its own
// text: written this way it took 11-17 spilled registers instead of 6 and 0.50 ms instead of 0.455 per 10 M reads.)  Lean build
// only: the dense build's walks over sites with alleles of any length and its allele windows are made for reads of up to 160
// bases.  What this build declines goes through the express pass' queue to the general pass (the express pass takes five k-mers).
// tests: every scenario with reads of 161..256 bases (tests/test_long_reads.py, tests/stress_emu.py) against the oracle, with
// correct, missing, shifted and foreign hints.
#pragma once
#include "hinted.hpp"

namespace gtx
{
// The build's geometry: NK k-mers, reads of up to MAX_READ bases, PLANE_WORDS words per bit plane, rows of ROW_BYTES bytes
template <uint32_t NK>
struct HintGeom
{
  static_assert(NK == 5 || NK == 8, "five k-mers (160 bases) or eight (256)");
  static constexpr uint32_t MAX_READ = NK == 5 ? 160u : 256u;
  static constexpr uint32_t PLANE_WORDS = MAX_READ / 32u; // words of a bit plane
  static constexpr uint32_t ROW_BYTES = MAX_READ / 2u;
};

// hint_compare's counters for NK k-mers: k as in HintCounts; upto = mismatches in [0, 31 j) for j = 1..4, upto2 for j = 5..8 (eight
// bits each); more = the whole read (bits 0..8) and the mismatch flags of the boundary bases 31 j, j = 1..NK-1 (bits 16..)
template <uint32_t NK>
struct HintCountsN
{
  uint32_t k[NK];
  uint32_t upto, more, upto2;
};

template <uint32_t NK>
GTX_DEV uint32_t hc_upto(HintCountsN<NK> const & h, uint32_t j)
{
  if constexpr (NK == 5)
    return j == 5 ? (h.more & 255u) : ((h.upto >> (8 * (j - 1))) & 255u);
  else
    return j > 4 ? ((h.upto2 >> (8 * (j - 5))) & 255u) : ((h.upto >> (8 * (j - 1))) & 255u);
}

template <uint32_t NK>
GTX_DEV uint32_t hc_all(HintCountsN<NK> const & h)
{
  if constexpr (NK == 5)
    return (h.more >> 8) & 255u;
  else
    return h.more & 511u;
}

template <uint32_t NK>
GTX_DEV uint32_t hc_edge(HintCountsN<NK> const & h, uint32_t j)
{
  return (h.more >> (15 + j)) & 1u;
}

template <uint32_t NK, class Row>
GTX_DEV void hint_compare_n(Row row, uint32_t seq_stride, uint32_t const * refp, uint32_t s, uint32_t L, HintCountsN<NK> & h)
{
  constexpr uint32_t PW = HintGeom<NK>::PLANE_WORDS;
  uint32_t mk[PW], am[PW], ao[PW], mt[PW + 1];
  // all loads first (no branches around them: the plane array is padded, a row has at least seq_stride bytes)
  // (the read's words come from the row -- LDS in the kernel -- group by group inside the loop: only the reference words,
  //  a global round trip, are worth holding all at once)
  uint32_t gg[4 * (PW + 1)];
#pragma unroll
  for (uint32_t w = 0; w < 4 * (PW + 1); ++w)
    gg[w] = refp[w];
#pragma unroll
  for (uint32_t W = 0; W < PW; ++W)
  {
    bool const in_row = 16 * W < seq_stride; // (uniform; a select, not a branch around the loads)
    uint32_t const at = in_row ? 4 * W : 0u, keep = in_row ? 0xFFFFFFFFu : 0u;
    uint32_t const r0 = row[at + 0] & keep, r1 = row[at + 1] & keep, r2 = row[at + 2] & keep, r3 = row[at + 3] & keep;
    uint32_t const g0 = hint_funnel(gg[4 * W + 0], gg[4 * W + 4], s), g1 = hint_funnel(gg[4 * W + 1], gg[4 * W + 5], s);
    uint32_t const g2 = hint_funnel(gg[4 * W + 2], gg[4 * W + 6], s), g3 = hint_funnel(gg[4 * W + 3], gg[4 * W + 7], s);
    uint32_t const v = L >= 32 * W + 32 ? 0xFFFFFFFFu : L <= 32 * W ? 0u : (1u << (L - 32 * W)) - 1u; // bases of the read
    uint32_t const differ = ((r0 ^ g0) | (r1 ^ g1) | (r2 ^ g2) | (r3 ^ g3)) & v;
    uint32_t const odd = r0 ^ r1 ^ r2 ^ r3, three = (r0 & r1 & (r2 | r3)) | (r2 & r3 & (r0 | r1));
    uint32_t const amb = ~(odd & ~three) & v;                              // not exactly one base: '=' (0), N, every other IUPAC set
    uint32_t const r_any = ((r0 & r1 & r2 & r3) | ~(r0 | r1 | r2 | r3)) & v; // N or '=' (which the reference reads as N)
    uint32_t const shares = (r0 & g0) | (r1 & g1) | (r2 & g2) | (r3 & g3);   // the read's set holds the reference base
    mk[W] = differ & ~amb;
    am[W] = amb;
    ao[W] = amb & ~(shares | r_any);
    mt[W] = differ & ~r_any & ~(g0 & g1 & g2 & g3); // count_mismatches (graph_utils.hpp:7-69)
    GTX_PIN(mk[W]);
    GTX_PIN(am[W]);
    GTX_PIN(ao[W]);
    GTX_PIN(mt[W]);
  }
  mt[PW] = 0;
  // ---- counters.  k-mer I is bases [31 I, 31 I + 32): a 32-bit window of the flag words (I = 0: word 0)
  uint32_t prefix = 0; // mismatches in the plane words in front of the current one
#pragma unroll
  for (uint32_t I = 0; I < NK; ++I)
  {
    uint32_t const W0 = ((K - 1) * I) / 32, o = ((K - 1) * I) % 32;
    uint32_t const nxt = W0 + 1 < PW ? W0 + 1 : W0; // (o = 0 only for I = 0: the next word is not looked at)
    uint32_t const wm = o == 0 ? mk[W0] : (mk[W0] >> o) | (mk[nxt] << (32 - o));
    uint32_t const wa = o == 0 ? am[W0] : (am[W0] >> o) | (am[nxt] << (32 - o));
    uint32_t const wo = o == 0 ? ao[W0] : (ao[W0] >> o) | (ao[nxt] << (32 - o));
    h.k[I] = static_cast<uint32_t>(__builtin_popcount(wm)) | (static_cast<uint32_t>(__builtin_popcount(wm & 0xFFFFu)) << HC_MIS_LEFT) |
             (static_cast<uint32_t>(__builtin_popcount(wa)) << HC_AMB) | (static_cast<uint32_t>(__builtin_popcount(wa & 0xFFFFu)) << HC_AMB_LEFT) |
             (static_cast<uint32_t>(__builtin_popcount(wo)) << HC_AMB_OUT);
    // mismatches by the walks' rule in [0, 31 (I + 1)): the words in front of word I + the low 31 - I bits of word I
    uint32_t const upto = prefix + static_cast<uint32_t>(__builtin_popcount(mt[I] & (0x7FFFFFFFu >> I)));
    if (I < 4)
      h.upto |= upto << (8 * I);
    else if (NK == 5)
      h.more |= upto;
    else
      h.upto2 |= upto << (8 * (I - 4));
    prefix += static_cast<uint32_t>(__builtin_popcount(mt[I]));
    if (I > 0) // the boundary base 31 I
      h.more |= ((mt[W0] >> o) & 1u) << (15 + I);
  }
#pragma unroll
  for (uint32_t W = NK; W < PW; ++W) // (none: as many plane words as k-mers)
    prefix += static_cast<uint32_t>(__builtin_popcount(mt[W]));
  h.more |= NK == 5 ? prefix << 8 : prefix; // the whole read
}

#if defined(__HIPCC__)
#define GTX_DEVF __device__ __forceinline__
#else
#define GTX_DEVF inline
#endif
// the k-th of NK values, k known at run time only (k >= FROM): k == FROM ? a[FROM] : k == FROM + 1 ? a[FROM + 1] : ... : a[NK - 1],
// a chain of selects over constant indices (written as a recursion, not as a loop: the optimiser recognises the loop for what
// it is, a[k], and the array then has to live in scratch)
template <uint32_t I, uint32_t NK>
GTX_DEVF uint32_t hint_sel_chain(uint32_t const (&a)[NK], uint32_t k)
{
  if constexpr (I + 1 < NK)
    return k == I ? a[I] : hint_sel_chain<I + 1, NK>(a, k);
  else
    return a[NK - 1];
}

template <uint32_t FROM, uint32_t NK>
GTX_DEVF uint32_t hint_sel(uint32_t const (&a)[NK], uint32_t k)
{
  return hint_sel_chain<FROM, NK>(a, k);
}

// every k-mer's verdict (hint_kmer<I> for I = 0 .. NK - 1; a k-mer the read does not have is a hole nobody looks at)
template <uint32_t I, uint32_t NK, class Row, class Counts>
GTX_DEVF void hint_kmers(uint32_t (&k)[NK], uint32_t const (&fx)[NK], uint32_t const (&fy)[NK], Row row, Counts const & h, uint32_t n_k, uint32_t none,
                        uint32_t & amb2)
{
  if constexpr (I < NK)
  {
    k[I] = (I < 2 || I < n_k) ? hint_kmer<I, false>(uint2_t{fx[I], fy[I]}, row, h, amb2) : none;
    hint_kmers<I + 1, NK>(k, fx, fy, row, h, n_k, none, amb2);
  }
}

// the filter slots of every k-mer's two halves (hint_probe_slot)
template <uint32_t I, uint32_t NK, class Row>
GTX_DEVF void hint_probe_slots(IndexView const & ix, uint32_t const (&k)[NK], Row row, uint32_t (&wl)[NK], uint32_t (&ml)[NK], uint32_t (&wr)[NK],
                              uint32_t (&mr)[NK])
{
  if constexpr (I < NK)
  {
    hint_probe_slot<I, 0>(ix, k[I], row, wl[I], ml[I]);
    hint_probe_slot<I, 1>(ix, k[I], row, wr[I], mr[I]);
    hint_probe_slots<I + 1, NK>(ix, k, row, wl, ml, wr, mr);
  }
}

template <uint32_t NK, class Row>
GTX_DEV uint32_t hinted_long_path(GraphView const & g, IndexView const & ix, Row row, uint32_t seq_stride, gtx_read_meta const & m, uint32_t idx,
                                  uint32_t * rec, uint32_t rec_words, uint32_t * stage)
{
  constexpr uint32_t PW = HintGeom<NK>::PLANE_WORDS;
  uint32_t const L = m.l_qseq;
  auto order_of = [&](uint32_t p) -> uint32_t { return g.first_order + p; };
  uint32_t const n_k = 1 + (L - K) / (K - 1);
  // ---- the read and the reference under it, 8 bases per word, aligned to the read
  HintCountsN<NK> h{};
  uint32_t const * refw = ix.refp + 4 * (idx >> 5);
  uint32_t const sh = idx & 31u;
  // ---- the flags of the k-mers' places (their second words also describe the positions the walks start from); all
  //      issued together with the reference words: one round trip
  uint32_t fx[NK], fy[NK];
#pragma unroll
  for (uint32_t i = 0; i < NK; ++i)
  {
    uint2_t const f = ix.pos_flags[idx + ((i < 2 || i < n_k) ? i * (K - 1) : 0u)];
    fx[i] = f.x;
    fy[i] = f.y;
  }
  uint32_t const y_end = ix.pos_flags[idx + (K - 1) * n_k].y; // the position behind the last k-mer (31 n_k <= L - 1: inside the read)
  uint2_t const t_end = ix.tail_info[idx + (K - 1) * n_k];    // ... and the site behind its reference node
  hint_compare_n<NK>(row, seq_stride, refw, sh, L, h);
  // ---- every k-mer: the label of its place, no label at all, or not provable
  uint32_t const none = hk_make(HINT_K_HOLE, HINT_NO_SITE, 0u, false, false);
  uint32_t amb2 = 0; // k-mers with one ambiguous base in each half: three more filter probes (below)
  uint32_t kv[NK];
  hint_kmers<0, NK>(kv, fx, fy, row, h, n_k, none, amb2);
  auto any_declined = [&]()
  {
    bool d = false;
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i)
      d = d || (kv[i] & 3u) == HINT_K_DECLINE;
    return d;
  };
  if (any_declined())
    return false;
  uint32_t k_any = 0;
#pragma unroll
  for (uint32_t i = 0; i < NK; ++i)
    k_any |= kv[i];
  if (k_any & (HK_NEED_LEFT | HK_NEED_RIGHT))
  {
    // ---- the filter probes of all k-mers together: one round trip
    uint32_t wl[NK], ml[NK], wr[NK], mr[NK], xl[NK], xr[NK];
    hint_probe_slots<0, NK>(ix, kv, row, wl, ml, wr, mr);
    uint32_t const * fl = ix.filt[0];
    uint32_t const * fr = ix.filt[1];
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i)
    {
      xl[i] = fl[wl[i]];
      xr[i] = fr[wr[i]];
    }
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i)
      kv[i] = hint_probe_verdict(kv[i], xl[i], ml[i], xr[i], mr[i]);
  }
  if (any_declined())
    return false;
  if (amb2 != 0)
  {
    // ---- a k-mer with one ambiguous base in each half (rare: one wavefront in fifteen meets one).  The three left halves
    //      that carry another base than the reference's at the ambiguous position must occur in no indexed key.  One such
    //      k-mer per read is looked at; a second one sends the read on.
    if ((amb2 & (amb2 - 1u)) != 0)
    {
      GTX_HINT_NOTE(7);
      return false;
    }
    uint32_t const A = (K - 1) * static_cast<uint32_t>(__builtin_ctz(amb2));
    uint32_t p[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b)
    {
      uint32_t const w = A >> 5, sft = A & 31u;
      uint32_t const lo_w = row[4 * w + b], hi_w = row[4 * (w + 1 < PW ? w + 1 : w) + b];
      p[b] = hint_funnel(lo_w, hi_w, sft) & 0xFFFFu; // bases A .. A+15: inside the row's groups
    }
    uint32_t const odd = p[0] ^ p[1] ^ p[2] ^ p[3], three = (p[0] & p[1] & (p[2] | p[3])) | (p[2] & p[3] & (p[0] | p[1]));
    uint32_t const amb16 = ~(odd & ~three) & 0xFFFFu;
    uint32_t const j = static_cast<uint32_t>(__builtin_ctz(amb16 | 0x10000u)) & 15u; // the ambiguous base of the left half
    uint32_t const q = idx + A + j;                                                    // ... and the reference base under it
    uint32_t const * rq = ix.refp + 4 * (q >> 5);
    uint32_t const qs = q & 31u;
    uint32_t const rc = ((rq[0] >> qs) & 1u) | (((rq[1] >> qs) & 1u) << 1) | (((rq[2] >> qs) & 1u) << 2) | (((rq[3] >> qs) & 1u) << 3);
    uint32_t const ref_two = rc == 1 ? 0u : rc == 2 ? 1u : rc == 4 ? 2u : 3u;
    uint32_t const lo0 = (p[1] | p[3]) & ~(1u << j), hi0 = (p[2] | p[3]) & ~(1u << j);
    bool maybe = (rc & (rc - 1u)) != 0 || (amb16 & (amb16 - 1u)) != 0; // (a reference N there, or not exactly one ambiguous base: not provable)
    uint32_t w3[3], m3[3];
#pragma unroll
    for (uint32_t t = 0; t < 3; ++t)
    {
      uint32_t const x = (ref_two + 1u + t) & 3u; // the three other bases
      hint_filter_slot(lo0 | ((x & 1u) << j), hi0 | ((x >> 1) << j), ix.filt_log2, w3[t], m3[t]);
    }
    uint32_t const * fl = ix.filt[0];
    uint32_t const y0 = fl[w3[0]], y1 = fl[w3[1]], y2 = fl[w3[2]];
    maybe = maybe || (y0 & m3[0]) == m3[0] || (y1 & m3[1]) == m3[1] || (y2 & m3[2]) == m3[2];
    if (maybe)
    {
      GTX_HINT_NOTE(4);
      return false;
    }
  }
  auto bits = [&](uint32_t flag, uint32_t want) // one bit per k-mer
  {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i)
      r |= (kv[i] & flag) == want ? 1u << i : 0u;
    return r;
  };
  uint32_t const labelled = bits(3u, HINT_K_LABEL), par = bits(HK_PAR, HK_PAR), mmk = bits(HK_MM, HK_MM);
  // ---- the run of k-mers that makes the path (express4.inl: the longest run of labelled k-mers, which has to be the
  //      only one of its length; the shorter side of a hole chains into a path remove_short_paths drops)
  uint32_t lo = 0, hi = n_k - 1;
  bool par_start = false;
  bool decided = false;                   // two runs of one length: the walks' outcome is worked out below, once for both
  uint32_t two_rs = 0, two_re = 0, two_mism = 0;
  if (labelled != (1u << n_k) - 1u)
  {
    uint32_t best_lo = 0, best_len = 0, second = 0, second_lo = 0, n_best = 0, cur_lo = 0, cur_len = 0;
#pragma unroll
    for (uint32_t kk = 0; kk <= NK; ++kk)
    {
      if (kk < n_k && ((labelled >> kk) & 1u))
      {
        cur_lo = cur_len == 0 ? kk : cur_lo;
        ++cur_len;
      }
      else
      {
        if (cur_len > best_len)
        {
          second = best_len;
          second_lo = best_lo;
          best_len = cur_len;
          best_lo = cur_lo;
          n_best = 1;
        }
        else if (cur_len != 0 && cur_len == best_len)
        {
          second = cur_len;
          second_lo = cur_lo;
          ++n_best;
        }
        else if (cur_len > second)
        {
          second = cur_len;
          second_lo = cur_lo;
        }
        cur_len = 0;
      }
    }
    if (best_len <= second)
    {
      // Two runs A (in front) and B of one length: both chains survive the first remove_short_paths and both are walked, A
      // first, with ONE shrinking budget per direction (genotype_paths.cpp:483-621: a walk that comes in under the best so
      // far drops the labels of the walks before it).  With the whole read inside one reference node every walk is one
      // compare with the linear reference -- the counts of hint_compare -- and the outcome is arithmetic: each chain's span
      // and mismatches behind the walks, the longer one stays (remove_short_paths).  Left to the general pass (which the
      // express pass would only hand it to as well): more than two such runs, a k-mer that opens a parallel chain, a site
      // under the read, chains that end up equally long (the reference returns both).
      uint32_t const a_lo = best_lo, b_lo = second_lo, len = best_len;
      uint32_t const run_a = ((1u << len) - 1u) << a_lo, run_b = ((1u << len) - 1u) << b_lo;
      if (best_len == 0 || n_best != 2 || (fy[0] & 255u) < L || (par & (run_a | run_b)) != 0)
      {
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      uint32_t const all = hc_all(h);
      auto upto = [&](uint32_t j) { return j == 0 ? 0u : hc_upto(h, j); };
      auto cap = [](uint32_t n, uint32_t best) { return 2u + n / 11u < best ? 2u + n / 11u : best; };
      // walk_read_starts: the bases [0, 31 lo] in front of (and with) each chain's first base
      uint32_t best = 7;
      uint32_t const head_a = upto(a_lo) + (a_lo ? hc_edge(h, a_lo) : 0u), head_b = upto(b_lo) + hc_edge(h, b_lo);
      bool got_a = a_lo != 0 && head_a <= cap((K - 1) * a_lo + 1u, best);
      best = got_a ? head_a : best;
      bool const got_b = head_b <= cap((K - 1) * b_lo + 1u, best);
      got_a = got_a && !(got_b && head_b < best); // (cannot be: A's piece is inside B's)
      uint32_t const rs_a = (got_a || a_lo == 0) ? 0u : (K - 1) * a_lo, rs_b = got_b ? 0u : (K - 1) * b_lo;
      uint32_t mm_a = static_cast<uint32_t>(__builtin_popcount(mmk & run_a)) + (got_a ? head_a : 0u);
      uint32_t mm_b = static_cast<uint32_t>(__builtin_popcount(mmk & run_b)) + (got_b ? head_b : 0u);
      // walk_read_ends: the bases [31 (hi + 1), L) behind (and with) each chain's last base
      uint32_t const end_a = (K - 1) * (a_lo + len), end_b = (K - 1) * (b_lo + len);
      uint32_t const tail_a = all - upto(a_lo + len), tail_b = all - upto(b_lo + len);
      best = 7;
      bool end_ok_a = tail_a <= cap(L - end_a, best);
      best = end_ok_a ? tail_a : best;
      bool const end_ok_b = end_b != L - 1 && tail_b <= cap(L - end_b, best);
      end_ok_a = end_ok_a && !(end_ok_b && tail_b < best); // (B's walk came in under A's: A's labels are dropped)
      uint32_t const re_a = end_ok_a ? L - 1 : end_a, re_b = (end_ok_b || end_b == L - 1) ? L - 1 : end_b;
      mm_a += end_ok_a ? tail_a : 0u;
      mm_b += end_ok_b ? tail_b : 0u;
      uint32_t const size_a = re_a - rs_a + 1u, size_b = re_b - rs_b + 1u;
      if (size_a == size_b)
      {
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      bool const a_wins = size_a > size_b;
      decided = true;
      two_rs = a_wins ? rs_a : rs_b;
      two_re = a_wins ? re_a : re_b;
      two_mism = a_wins ? mm_a : mm_b;
      best_lo = a_wins ? a_lo : b_lo;
    }
    // A run that opens, behind a label-less k-mer, with a k-mer that brings TWO label lists (a multi-key list is added with
    // 0 and with 1 mismatch, alignment.cpp:57-63; an exact key with indexed neighbours has its own and theirs) starts two
    // parallel chains: see `twin` below.
    par_start = !decided && best_lo > 0 && ((par >> best_lo) & 1u) != 0;
    lo = best_lo;
    hi = best_lo + best_len - 1;
  }
  uint32_t const run = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
  uint32_t mism = static_cast<uint32_t>(__builtin_popcount(mmk & run));
  // ---- the read in front of the run and behind it: the walks' shortcut, both inside the reference node the path touches
  uint32_t const prs = (K - 1) * lo, pre = (K - 1) * (hi + 1);
  uint32_t start = order_of(idx + prs), rs = prs;
  // A walk that leaves its reference node over ONE site whose alleles are single bases (tail_info: HINT_TAIL_OK), with the
  // rest inside the reference node on the other side: Graph::get_labels_forward / _backward has one candidate per allele,
  // they differ in that character only, and the labels of the best ones share their ends -- one path with the site's best
  // alleles (express4.inl).  The reference allele and both nodes ARE the linear reference, so the compare above already
  // holds every other character.  `only` < 4: the walk starts INSIDE that allele (the path ends on the site's base and
  // carries the allele: Graph::get_locations_of_a_position offers variant nodes the path has, graph.cpp:1154-1185).
  // rc: the read's base on the site.  Returns the mismatches there for the best alleles (their set in `mask`) and what
  // the compare with the reference allele had counted (x0).
  auto site_choice = [&](uint32_t tx, uint32_t rc0, uint32_t only, uint32_t & mask, uint32_t & x0) -> uint32_t
  {
    uint32_t const rc = rc0 == 0 ? 15u : rc0; // ('=' reads as N)
    uint32_t const nall = (tx >> HINT_TAIL_NALL_SHIFT) & 7u, codes = tx >> HINT_TAIL_CODES_SHIFT;
    uint32_t best = 2;
    mask = 0;
#pragma unroll
    for (uint32_t a = 0; a < 4; ++a)
      if (a < nall && (only >= 4 || a == only))
      {
        uint32_t const gc = (codes >> (4 * a)) & 15u;
        uint32_t const xa = (gc != rc && rc != 15u) ? 1u : 0u; // (the alleles are A, C, G or T)
        if (xa < best)
        {
          best = xa;
          mask = 0;
        }
        if (xa == best)
          mask |= 1u << a;
      }
    x0 = ((codes & 15u) != rc && rc != 15u) ? 1u : 0u;
    return best;
  };
  // the single allele k-mer `km` carries on `site` (4: it is another site, or a set of several)
  auto carried = [&](uint32_t km, uint32_t site) -> uint32_t
  {
    uint32_t const set = (km >> HK_SET_SHIFT) & 255u;
    if ((km >> HK_SITE_SHIFT) != site || (set & (set - 1u)) != 0)
      return 4u;
    return set ? static_cast<uint32_t>(__builtin_ctz(set)) : (km >> HK_ALLELE_SHIFT) & 3u;
  };
  uint32_t head_site = 0, head_mask = 0; // the site the walk at the read's start crossed, with its best alleles
  if (prs != 0 && !decided) // walk_read_starts (genotype_paths.cpp:555-621)
  {
    uint32_t const y = hint_sel<1>(fy, lo); // (position 31 lo is k-mer lo's own place)
    uint32_t const back = (y >> HINT_BACK_SHIFT) & 255u;
    uint32_t upto = hc_upto(h, lo) + hc_edge(h, lo); // mismatches in [0, prs]: the boundary base itself is base 31 lo
    if ((y & 255u) == 0 || back < prs)
    {
      // the walk leaves the node backwards: over the site in front of it (its base is read base ps), or -- the path
      // starts ON a site's base -- out of the allele it carries into the node in front
      bool const on_site = (y & 255u) == 0;
      uint32_t const ps = on_site ? prs : prs - back - 1u;
      if (idx + ps == 0)
      {
        GTX_HINT_NOTE(11);
        return false;
      }
      uint32_t const q = idx + ps - 1u; // the position in front of the site: the last base of the node there
      uint32_t const yq = ix.pos_flags[q].y;
      uint2_t const tq = ix.tail_info[q];
      uint32_t const km = hint_sel<1>(kv, lo);
      uint32_t const only = on_site ? carried(km, tq.y) : 4u;
      if ((yq & 255u) != 1u || (tq.x & HINT_TAIL_OK) == 0 || (ps != 0 && ((yq >> HINT_BACK_SHIFT) & 255u) + 1u < ps) || (on_site && only >= 4u))
      {
        GTX_HINT_NOTE(11);
        return false; // (an indel, a second site, a set of alleles: express4 / general pass)
      }
      uint32_t mask = 0, x0 = 0;
      uint32_t const best = site_choice(tq.x, plane_code_at(row, ps), only, mask, x0);
      upto = upto - x0 + best;
      head_site = tq.y;
      head_mask = mask;
    }
    uint32_t const head_len = prs + 1;
    uint32_t const budget = 2 + head_len / 11 < 7 ? 2 + head_len / 11 : 7; // genotype_paths.cpp:571-577
    if (upto <= budget)
    {
      start = order_of(idx);
      rs = 0;
      mism += upto;
    }
    else
      head_mask = 0; // (the path stays as it is: no site from the walk)
  }
  uint32_t end = order_of(idx + pre), re = pre;
  uint32_t tail_site = 0, tail_mask = 0; // the site the walk at the read's end crossed, with its best alleles
  if (decided) // (two runs of one length, inside one reference node: worked out above)
  {
    start = order_of(idx + two_rs);
    rs = two_rs;
    re = two_re;
    end = order_of(idx + two_re);
    mism = two_mism;
  }
  else if (pre != L - 1) // walk_read_ends (genotype_paths.cpp:483-553)
  {
    uint32_t const tail_len = L - pre;
    uint32_t const budget = 2 + tail_len / 11 < 7 ? 2 + tail_len / 11 : 7; // genotype_paths.cpp:505-511
    uint32_t tail_end = order_of(idx + L - 1u); // (inside one reference node, or over SNP-like sites: the path's own position)
    uint32_t const y = hi + 1 == n_k ? y_end : hint_sel<1>(fy, hi + 1);
    uint32_t const room = y & 255u;
    uint32_t got = hc_all(h) - hc_upto(h, hi + 1);
    if (room < tail_len)
    {
      // the tail leaves the node: over the site behind it (tail character `at`), or -- the path ends ON a site's base --
      // out of the allele it carries into the node behind
      bool const on_site = room == 0;
      uint32_t const at = room;
      uint2_t ti = t_end;
      uint32_t only = 4u;
      bool ok = true;
      if (on_site)
      {
        uint32_t const q = idx + pre - 1u; // (pre >= 32) the last base of the node in front of the site
        ok = (ix.pos_flags[q].y & 255u) == 1u;
        ti = ix.tail_info[q];
        only = carried(hint_sel<0>(kv, hi), ti.y);
        ok = ok && only < 4u;
      }
      else if (hi + 1 != n_k)
        ti = ix.tail_info[idx + pre]; // (the run ends in front of a label-less k-mer: the table entry of that place)
      uint32_t const next_len = (ti.x >> HINT_TAIL_NEXT_SHIFT) & 255u;
      if (!ok || (ti.x & HINT_TAIL_OK) == 0 || next_len < tail_len - at - 1)
      {
        // (an indel, a second site, a set of alleles.  The dense build walks over up to two sites with alleles of any length
        //  itself; what it cannot decide there -- best candidates with different ends: several paths -- the express pass
        //  cannot either.  Lean build: express4 / general pass)
        GTX_HINT_NOTE(8);
        return false;
      }
      else
      {
        uint32_t mask = 0, x0 = 0;
        uint32_t const best = site_choice(ti.x, plane_code_at(row, pre + at), only, mask, x0);
        got = got - x0 + best;
        tail_site = ti.y;
        tail_mask = mask;
      }
    }
    if (got <= budget)
    {
      re = L - 1;
      mism += got;
      end = tail_end;
    }
    else
      tail_mask = 0; // (the path stays as it is: no site from the walk)
  }
  // ---- variant sites of the path, most recent k-mer first (Path(p1, p2), path.cpp:38-82); a site under two
  //      neighbouring k-mers is one entry (the same base, hence the same allele)
  uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0; // site << 16 | allele mask (named registers: an indexed array would live in scratch)
  uint32_t nvar = 0;
  uint32_t last = 0xFFFFFFFFu;
  bool clash = false;
  auto append = [&](uint32_t entry)
  {
    if ((entry >> 16) == (last >> 16))
    {
      // the site again (under the neighbouring k-mer, or the walk's): the allele sets are intersected (path.cpp:38-82)
      last &= entry | 0xFFFF0000u;
      clash = clash || (last & 0xFFFFu) == 0;
      v0 = nvar == 1 ? last : v0;
      v1 = nvar == 2 ? last : v1;
      v2 = nvar == 3 ? last : v2;
      v3 = nvar == 4 ? last : v3;
      v4 = nvar == 5 ? last : v4;
      v5 = nvar == 6 ? last : v5;
    }
    else
    {
      v0 = nvar == 0 ? entry : v0;
      v1 = nvar == 1 ? entry : v1;
      v2 = nvar == 2 ? entry : v2;
      v3 = nvar == 3 ? entry : v3;
      v4 = nvar == 4 ? entry : v4;
      v5 = nvar == 5 ? entry : v5;
      ++nvar;
      last = entry;
    }
  };
  auto push = [&](uint32_t k, uint32_t km)
  {
    if (((run >> k) & 1u) && (km >> HK_SITE_SHIFT) != HINT_NO_SITE)
    {
      uint32_t const set = (km >> HK_SET_SHIFT) & 255u;
      append(((km >> HK_SITE_SHIFT) << 16) | (set ? set : 1u << ((km >> HK_ALLELE_SHIFT) & 3u)));
    }
  };
  if (tail_mask != 0) // (the walk's labels are merged last: their site comes first)
    append((tail_site << 16) | tail_mask);
#pragma unroll
  for (uint32_t j = 0; j < NK; ++j)
    push(NK - 1u - j, kv[NK - 1u - j]);
  if (head_mask != 0) // (Path(pp, original), genotype_paths.cpp:262: the original's sites stay in front, the start walk's come last)
    append((head_site << 16) | head_mask);
  if (clash || nvar > 6 || 6 + 3 * nvar > rec_words) // (six named registers hold the sites)
  {
    GTX_HINT_NOTE(13);
    return false;
  }
  uint32_t np = 1, longest = re - rs + 1;
  if (par_start && rs == 0)
  {
    // Two parallel chains P0 (m mismatches) and P1 (m + 1) opened the run, and the walk at the read's start succeeded for
    // both.  walk_read_starts then holds the same label list twice (genotype_paths.cpp:596-612): the first copy extends P0
    // and P1, the second finds no path left to merge with and becomes a path D of its own over read bases 0 .. 31 lo
    // (add_prev_kmer_labels, :282-290).  walk_read_ends visits P0, P1, D in that order with a shrinking budget (:497-531):
    // D's walk runs over the k-mers AND the tail and has to come in at the tail's own mismatch count -- possible exactly
    // when no k-mer of the run took its label from a Hamming-1 list.  D then grows into a full-length twin of P0 with the
    // same mismatches, survives remove_paths_with_too_many_mismatches beside it (P1 does not), and the reference really
    // returns the path twice.  With at most one site the twin's site list is P0's (the same walks, the same best alleles).
    // Left to the general pass: a read without a tail (D is then measured against 7, not against the tail), a failed tail
    // walk (D may overtake P0 alone), twins over several sites (the order of a walk's sites is the walk's business).
    // The two chains walk alike unless the walk starts INSIDE a variant node and the second chain is another allele's (an
    // exact k-mer whose neighbours are the site's other alleles: its walk starts with a mismatch, its list is not kept, no
    // D): a multi-key list (an ambiguity code in the k-mer) gives the same labels twice wherever it starts.
    // ... and the second chain has to live as long as the first: another allele's chain dies where the next k-mer names the
    // site again (a SNP on the k-mer's last base is the next k-mer's first: Path(p1, p2) finds no allele in common).
    uint32_t const amb_lo = hc_get(hint_sel<1>(h.k, lo), HC_AMB);
    uint2_t const f_lo{hint_sel<1>(fx, lo), hint_sel<1>(fy, lo)};
    bool two_chains = amb_lo != 0;
    if (!two_chains)
    {
      uint32_t const off = (f_lo.y >> HINT_SNPOFF_SHIFT) & 31u;
      if ((f_lo.x & HINT_ALT_OK) == 0 || off == K - 1)
      {
        // (neighbours that are not a SNP's alleles; a SNP on the k-mer's last base -- the next k-mer's first: whether the
        //  other allele's chain lives on depends on what kind of list that k-mer brings)
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      two_chains = off != 0;
    }
    if (two_chains)
    {
      // D reaches the read's end at the tail's mismatch count when bases 31 lo .. pre - 1 hold none: no k-mer of the run
      // took its label from a Hamming-1 list -- or only the last one did, for a substitution on its last base, which is the
      // tail walk's first (the chain counted it twice, the twin counts it once and is returned ALONE, one mismatch less)
      // (counted by the walks' rule -- an ambiguity code that is not N is a character of its own there, whatever its k-mer's
      //  lists found -- against the linear reference, which is what the walk sees as long as the run carries reference
      //  alleles only: a run with another allele or a set of alleles is left to the general pass)
      uint32_t const mm_run = mmk & run, km_hi = hint_sel<0>(kv, hi);
      bool all_plain = true;
#pragma unroll
      for (uint32_t i = 0; i < NK; ++i)
        all_plain = all_plain && (((run >> i) & 1u) == 0 || (kv[i] & ((3u << HK_ALLELE_SHIFT) | (255u << HK_SET_SHIFT))) == 0);
      if (!all_plain)
      {
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      bool const region_clean = hc_upto(h, hi + 1) == hc_upto(h, lo);
      if (mm_run == (1u << hi) && hi + 1 > NK - 1)
      {
        // (the build's last k-mer -- the fifth of a read of 156 bases and more: whether its substitution sits on its last base --
        //  base 155, the tail walk's first -- is not among the compare's counts)
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      bool const last_only = mm_run == (1u << hi) && hc_edge(h, hi + 1) == 1 && ((km_hi >> HK_SET_SHIFT) & 255u) == 0;
      bool const twin = region_clean && (mm_run == 0 || last_only);
      if (pre == L - 1 || re != L - 1 || (twin && nvar > 1))
      {
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      if (twin && mm_run == 0)
        np = 2;
      else if (twin)
        --mism;
    }
  }
  if (mism > 10) // remove_paths_with_too_many_mismatches
  {
    np = 0;
    longest = 0;
  }
  uint32_t const path_words = 4 + 3 * nvar;
  if (2 + np * path_words > rec_words)
    return false;
  bool const to_stage = stage != nullptr && rec_words >= HINT_STAGE_WORDS && 2 + (np ? np : 1u) * path_words <= HINT_STAGE_WORDS;
  if (to_stage)
  {
    rec = stage;
#pragma unroll
    for (uint32_t k = 2; k < HINT_STAGE_WORDS; ++k)
      rec[k] = 0u;
  }
  rec[0] = np;
  rec[1] = longest | (L << 16) | ((np && nvar) ? GTX_REC_HAS_VARIANTS : 0u);
  if (np)
  {
    rec[2] = start;
    rec[3] = end;
    rec[4] = rs | (re << 16);
    rec[5] = mism | (nvar << 16);
    auto put = [&](uint32_t k, uint32_t entry)
    {
      if (k < nvar)
      {
        rec[6 + 3 * k] = entry >> 16;
        rec[7 + 3 * k] = entry & 0xFFFFu;
        rec[8 + 3 * k] = 0u;
      }
    };
    put(0, v0);
    put(1, v1);
    put(2, v2);
    put(3, v3);
    put(4, v4);
    put(5, v5);
    if (np == 2) // the twin: the same words again
      for (uint32_t k = 0; k < path_words; ++k)
        rec[2 + path_words + k] = rec[2 + k];
  }
  return to_stage ? 2u : 1u;
}


// The forward task of one read of up to HintGeom<NK>::MAX_READ bases (hinted_one's contract)
template <uint32_t NK, class Row>
GTX_DEV uint32_t hinted_long_one(GraphView const & g, IndexView const & ix, Row row, uint32_t seq_stride, gtx_read_meta const & m, uint32_t * rec,
                                 uint32_t rec_words, uint32_t * stage = nullptr)
{
  uint32_t const L = m.l_qseq;
  if (L < 2 * K - 1 || L > HintGeom<NK>::MAX_READ || m.pos < 0 || ix.n_hint == 0)
  {
    GTX_HINT_NOTE(9);
    return false;
  }
  if (static_cast<uint32_t>(m.pos) < ix.hint_first)
    return false;
  uint32_t const idx = static_cast<uint32_t>(m.pos) - ix.hint_first;
  if (idx >= ix.n_hint || L > ix.n_hint - idx)
    return false;
  return hinted_long_path<NK>(g, ix, row, seq_stride, m, idx, rec, rec_words, stage);
}

} // namespace gtx
