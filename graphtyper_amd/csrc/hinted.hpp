// hinted.hpp -- pass 0 of the alignment: ONE READ PER LANE, guided by the position the read comes with.
//
// The reference looks every k-mer of a read up globally -- the exact key and its 96 Hamming-1 neighbours
// (src/utilities/kmer_help_functions.cpp:53-119, src/index/ph_index.cpp:66-107) -- because it does not use where the
// mapper put the read.  A record of a sorted BAM does carry that place (bam1_t::core.pos, handed over as
// gtx_read_meta::pos).  This pass compares the read with the linear reference AT that place, bit-parallel on packed
// nibbles, and takes the answers of the global lookups from flags computed once per reference position when the index
// was built (IndexView::pos_flags, gtx_host.cpp: build_hints) plus, for a k-mer with one substitution, one probe of a
// half-key presence filter.  It finishes a read only when every lookup of the reference is PROVEN to return the one
// label of that place:
//
//   k-mer == reference 32-mer K          HINT_EXACT_OK: K has that one label and all its indexed Hamming-1 neighbours
//                                        are the same interval on the same site (they end where the chain ends);
//   one substitution against K           the half without it is shared with K only (HINT_L1 / HINT_R1: no other
//                                        neighbour there), the half with it occurs in no indexed key (filter bit clear:
//                                        no exact hit, no neighbour there) -> the Hamming-1 list is K's label, +1 mismatch;
//   one ambiguous base, rest == K        the keys of its expansion differ from K in that base only, so they share the
//                                        other half with K, which K alone has -> of the expansion only K is indexed.
//
// With every k-mer settled that way the read is the "simple read" of express4.inl (one label per k-mer, abutting by
// construction) and gets the record express4 would write: same start / end / mismatches / sites.  The rest of the read
// behind the last k-mer must lie inside the reference node (flags again) and is counted against the same nibbles.
// Anything else -- a missing or wrong hint, a k-mer with two differences, a variant allele, a tail over a site, a long
// read -- is DECLINED untouched and goes to the global-lookup passes (express4, general, HBM tables), which do not look
// at the hint.  So the hint can only decide who does the work, never the result (tests feed wrong / shifted / missing
// hints and compare every record with the oracle).
//
// Per read: 20 B meta + 80 B bases + ~84 B of reference nibbles + 6 flag words, all but the bases shared with the
// neighbouring reads of the sorted stream; no hash probe at all for an error-free read.  Included from align_core.hpp.
#pragma once
#include "graph_dev.hpp"

namespace gtx
{
#ifdef GTX_EMU_NOTES // diagnostics of the host emulation (tests/emu): why a read leaves this pass
void hint_note(uint32_t code);
#define GTX_HINT_NOTE(code) hint_note(code)
#else
#define GTX_HINT_NOTE(code) ((void)0)
#endif

constexpr uint32_t HINT_MAX_READ = 160; // bases (20 words); longer reads are left to express4
constexpr uint32_t HINT_WORDS = HINT_MAX_READ / 8;

GTX_DEV uint32_t hint_bswap(uint32_t v)
{
  return __builtin_bswap32(v);
}

// one flag bit per nibble (at the nibble's lowest bit)
GTX_DEV uint32_t nib_nonzero(uint32_t v)
{
  v |= v >> 1;
  v |= v >> 2;
  return v & 0x11111111u;
}

GTX_DEV uint32_t nib_is15(uint32_t v)
{
  v &= v >> 1;
  v &= v >> 2;
  return v & 0x11111111u;
}

// nibble is not exactly one of 1, 2, 4, 8
GTX_DEV uint32_t nib_not_onehot(uint32_t v)
{
  // per nibble popcount (0..4) in place, then "!= 1"
  uint32_t c = v - ((v >> 1) & 0x77777777u);
  c -= (v >> 2) & 0x33333333u;
  c -= (v >> 3) & 0x11111111u;
  return nib_nonzero(c ^ 0x11111111u);
}

// flag bits of the nibbles [a, b) of a 160-nibble string held as HINT_WORDS words (nibble 8w+j at bits 28-4j of word w)
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr uint32_t nib_range_mask(uint32_t w, uint32_t a, uint32_t b)
{
  uint32_t const lo = a > 8 * w ? a - 8 * w : 0u, hi = b > 8 * w ? b - 8 * w : 0u; // nibbles [lo, hi) of this word, clamped below
  uint32_t const from = lo >= 8 ? 0u : 0x11111111u >> (4 * lo); // nibbles lo..7
  uint32_t const upto = hi >= 8 ? 0u : 0x11111111u >> (4 * hi); // nibbles hi..7
  return from & ~upto;
}

// 16 nibbles from nibble A on, as two words (A is a compile-time constant)
template <uint32_t A>
GTX_DEV void nib_extract16(uint32_t const (&r)[HINT_WORDS], uint32_t & w0, uint32_t & w1)
{
  constexpr uint32_t W = A / 8, S = 4 * (A % 8);
  if constexpr (S == 0)
  {
    w0 = r[W];
    w1 = r[W + 1];
  }
  else
  {
    w0 = (r[W] << S) | (r[W + 1] >> (32 - S));
    w1 = (r[W + 1] << S) | ((W + 2 < HINT_WORDS ? r[W + 2] : 0u) >> (32 - S));
  }
}

// What the compare of the read with the reference under it says, summed while the words stream by (nothing but these
// counters and the read's own words stays in registers):
struct HintCounts
{
  uint32_t mis[AlignCfg::KC], mis_left[AlignCfg::KC]; // per k-mer: substitutions among its unambiguous bases (left = 16 first)
  uint32_t amb[AlignCfg::KC], amb_left[AlignCfg::KC]; // ... ambiguous bases
  uint32_t amb_out[AlignCfg::KC];                     // ... ambiguous bases whose set does not hold the reference base
  uint32_t upto[AlignCfg::KC + 1]; // mismatches (the walks' rule) in [0, 31 j); upto[0]: in the whole read
  uint32_t edge[AlignCfg::KC];     // ... at base 31 j itself
};

template <uint32_t W, uint32_t I>
GTX_DEV void hint_count_kmer(uint32_t mk, uint32_t am, uint32_t ao, uint32_t mt, HintCounts & h)
{
  constexpr uint32_t A = (K - 1) * I;
  constexpr uint32_t M = nib_range_mask(W, A, A + 32), ML = nib_range_mask(W, A, A + 16);
  constexpr uint32_t C = nib_range_mask(W, 0, A + (K - 1)), E = I == 0 ? 0u : nib_range_mask(W, A, A + 1);
  if constexpr (M != 0)
  {
    h.mis[I] += static_cast<uint32_t>(__builtin_popcount(mk & M));
    h.amb[I] += static_cast<uint32_t>(__builtin_popcount(am & M));
    h.amb_out[I] += static_cast<uint32_t>(__builtin_popcount(ao & M));
  }
  if constexpr (ML != 0)
  {
    h.mis_left[I] += static_cast<uint32_t>(__builtin_popcount(mk & ML));
    h.amb_left[I] += static_cast<uint32_t>(__builtin_popcount(am & ML));
  }
  if constexpr (C != 0)
    h.upto[I + 1] += static_cast<uint32_t>(__builtin_popcount(mt & C));
  if constexpr (E != 0)
    h.edge[I] += static_cast<uint32_t>(__builtin_popcount(mt & E));
}

// word W of the read (rw: 8 bases, base j in bits 28-4j; `have` bases of the read lie in it) against the reference (gw)
template <uint32_t W>
GTX_DEV uint32_t hint_word(uint32_t rw, uint32_t gw, uint32_t have, HintCounts & h)
{
  uint32_t const keep = have >= 8 ? 0xFFFFFFFFu : have == 0 ? 0u : ~(0xFFFFFFFFu >> (4 * have));
  rw &= keep;
  gw &= keep;
  uint32_t const flags = keep & 0x11111111u;
  uint32_t const differ = nib_nonzero(rw ^ gw);
  uint32_t const am = nib_not_onehot(rw) & flags;                     // '=' (0), N and every other IUPAC set
  uint32_t const r_any = (nib_is15(rw) | (~nib_nonzero(rw) & flags)); // N or '=' (which the reference reads as N)
  uint32_t const mk = differ & ~am;
  uint32_t const ao = am & ~(nib_nonzero(rw & gw) | r_any);           // the set misses the reference base
  uint32_t const mt = differ & ~r_any & ~nib_is15(gw);                // count_mismatches (graph_utils.hpp:7-69)
  hint_count_kmer<W, 0>(mk, am, ao, mt, h);
  hint_count_kmer<W, 1>(mk, am, ao, mt, h);
  hint_count_kmer<W, 2>(mk, am, ao, mt, h);
  hint_count_kmer<W, 3>(mk, am, ao, mt, h);
  hint_count_kmer<W, 4>(mk, am, ao, mt, h);
  h.upto[0] += static_cast<uint32_t>(__builtin_popcount(mt));
  return rw;
}

template <uint32_t W, class Row>
GTX_DEV void hint_compare_from(Row row, uint32_t seq_stride, uint32_t const * refw, uint32_t sh, uint32_t L, uint32_t prev, uint32_t (&r)[HINT_WORDS],
                               HintCounts & h)
{
  if constexpr (W < HINT_WORDS)
  {
    uint32_t const have = L > 8 * W ? L - 8 * W : 0u; // bases of the read in this word
    uint32_t rw = 0, gw = 0, next = 0;
    if (have != 0)
    {
      if (4 * W < seq_stride)
        rw = hint_bswap(row[W]);
      next = refw[W + 1];
      gw = sh == 0 ? prev : ((prev << sh) | (next >> (32 - sh)));
    }
    r[W] = hint_word<W>(rw, gw, have, h);
    hint_compare_from<W + 1>(row, seq_stride, refw, sh, L, next, r, h);
  }
}

template <class Row>
GTX_DEV void hint_compare(Row row, uint32_t seq_stride, uint32_t const * refw, uint32_t sh, uint32_t L, uint32_t (&r)[HINT_WORDS], HintCounts & h)
{
  hint_compare_from<0>(row, seq_stride, refw, sh, L, refw[0], r, h);
}

// the half's bits of the blocked Bloom filter are all set: an indexed key MAY have these 16 bases
GTX_DEV bool hint_half_maybe(IndexView const & ix, uint32_t side, uint32_t w0, uint32_t w1)
{
  uint32_t word, mask;
  hint_filter_slot(w0, w1, ix.filt_log2, word, mask);
  return (ix.filt[side][word] & mask) == mask;
}

enum : uint32_t
{
  HINT_K_DECLINE = 0,
  HINT_K_LABEL = 1, // the k-mer's lists hold exactly the label of this place
  HINT_K_HOLE = 2   // the k-mer has no label at all
};

struct HintKmer
{
  uint32_t kind;
  uint32_t site, allele; // of the label (site HINT_NO_SITE: none)
  bool mm;               // the label comes from the Hamming-1 list: one more mismatch
  bool par;              // the k-mer also starts a parallel chain (matters when it opens the run behind a hole)
};

// What the global lookups of the reference return for k-mer I of the read, proven from the flags of the hinted place
// (cases in the file header; a hole needs both halves of the k-mer to occur in no indexed key, or to be K's own halves
// while the k-mer is two or more substitutions away from K).
template <uint32_t I>
GTX_DEV HintKmer hint_kmer(IndexView const & ix, uint32_t idx, uint8_t const * seq4, uint32_t const (&r)[HINT_WORDS], HintCounts const & h)
{
  constexpr uint32_t A = (K - 1) * I;
  uint2_t const f = ix.pos_flags[idx + A];
  uint32_t const mis = h.mis[I], mis_left = h.mis_left[I], mis_right = mis - mis_left;
  uint32_t const amb = h.amb[I], amb_left = h.amb_left[I];
  uint32_t const amb_out = h.amb_out[I]; // ambiguous bases whose set does not hold the reference base
  HintKmer k{HINT_K_DECLINE, f.x >> HINT_SITE_SHIFT, 0u, false, false};
  bool const single = (f.x & HINT_SINGLE_OK) != 0, l1 = single && (f.x & HINT_L1) != 0, r1 = single && (f.x & HINT_R1) != 0;
  if (amb == 0 && mis == 0)
  {
    k.kind = (f.x & HINT_EXACT_OK) ? HINT_K_LABEL : HINT_K_DECLINE;
    k.par = (f.x & HINT_PAR) != 0;
    GTX_HINT_NOTE(k.kind == HINT_K_DECLINE ? 1 : 0); // exact k-mer, but the place is not provably simple
    return k;
  }
  if (amb > 1)
  {
    GTX_HINT_NOTE(7);
    return k;
  }
  if (amb == 0 && mis == 1 && (f.x & HINT_ALT_OK) != 0)
  {
    // the other allele of the SNP under the k-mer?  (one difference, and the base on the site is an alternative allele)
    uint32_t const at = A + ((f.y >> HINT_SNPOFF_SHIFT) & 31u);
    uint32_t const rb = (seq4[at >> 1] >> ((~at & 1u) << 2)) & 15u;
    uint32_t const two = rb == 1 ? 0u : rb == 2 ? 1u : rb == 4 ? 2u : 3u;
    uint32_t const allele = (f.x >> (HINT_ALTIDX_SHIFT + 2 * two)) & 3u;
    if (allele != 0)
    {
      k.kind = HINT_K_LABEL;
      k.allele = allele;
      k.par = true; // (the reference allele's key is its neighbour)
      return k;
    }
  }
  // the k-mer is not K: which of its halves are K's, which must be shown to occur in no indexed key
  uint32_t l0, l1w, r0, r1w;
  nib_extract16<A>(r, l0, l1w);
  nib_extract16<A + 16>(r, r0, r1w);
  if (amb == 0)
  {
    if (mis == 1)
    {
      bool const left = mis_left == 1;
      if (!(left ? r1 : l1))
      {
        GTX_HINT_NOTE(3); // one substitution, the other half is shared with further keys (a variant there)
        return k;
      }
      if (hint_half_maybe(ix, left ? 0u : 1u, left ? l0 : r0, left ? l1w : r1w))
      {
        GTX_HINT_NOTE(4); // one substitution, its half may occur in the index (a variant allele, or a filter collision)
        return k;
      }
      k.kind = HINT_K_LABEL;
      k.mm = true;
      return k;
    }
    // two or more substitutions: no label at all when neither half leads to an indexed key within distance 1 -- a half
    // without a substitution is K's own (K alone must have it: K itself is too far away), a half with one must occur in
    // no indexed key
    bool const ok = (mis_left == 0 ? l1 : !hint_half_maybe(ix, 0u, l0, l1w)) && (mis_right == 0 ? r1 : !hint_half_maybe(ix, 1u, r0, r1w));
    k.kind = ok ? HINT_K_HOLE : HINT_K_DECLINE;
    GTX_HINT_NOTE(ok ? 0 : 6);
    return k;
  }
  // one ambiguous base (a multi-key list: exact lookups only, src/utilities/kmer_help_functions.cpp:97-119)
  bool const amb_is_left = amb_left == 1;
  if (mis == 0)
  {
    // its keys differ from K in that base only: they share the other half with K
    if (!(amb_is_left ? r1 : l1))
    {
      GTX_HINT_NOTE(5);
      return k;
    }
    k.kind = amb_out == 0 ? HINT_K_LABEL : HINT_K_HOLE; // (a set without the reference base: none of its keys is K)
    k.par = true;
    return k;
  }
  // ... plus substitutions: none of its keys is K.  Either everything lies in one half (the other one is K's), or the
  // substitutions lie in the half without the ambiguous base, which then is one concrete 16-mer
  bool ok = false;
  if (amb_is_left)
    ok = mis_right == 0 ? r1 : (mis_left == 0 && !hint_half_maybe(ix, 1u, r0, r1w));
  else
    ok = mis_left == 0 ? l1 : (mis_right == 0 && !hint_half_maybe(ix, 0u, l0, l1w));
  k.kind = ok ? HINT_K_HOLE : HINT_K_DECLINE;
  k.par = true;
  GTX_HINT_NOTE(ok ? 0 : 7);
  return k;
}

// The forward task of one read.  Returns true when the record was written, false = declined (nothing written).
// `row`: the read's packed bases as words (global memory, or the copy the kernel staged in LDS); seq4 = the same bytes.
template <class Row>
GTX_DEV bool hinted_one(GraphView const & g, IndexView const & ix, Row row, uint8_t const * seq4, uint32_t seq_stride, gtx_read_meta const & m,
                        uint32_t * rec, uint32_t rec_words)
{
  uint32_t const L = m.l_qseq;
  if (L < 2 * K - 1 || L > HINT_MAX_READ || m.pos < 0 || ix.n_hint == 0)
  {
    GTX_HINT_NOTE(9);
    return false;
  }
  // position of read base 0 in the hint tables
  if (static_cast<uint32_t>(m.pos) < ix.hint_first)
    return false;
  uint32_t const idx = static_cast<uint32_t>(m.pos) - ix.hint_first;
  if (idx >= ix.n_hint || L > ix.n_hint - idx)
    return false;
  uint32_t const n_k = 1 + (L - K) / (K - 1);
  // ---- the read and the reference under it, 8 bases per word, aligned to the read
  uint32_t r[HINT_WORDS];
  HintCounts h{};
  uint32_t const * refw = ix.ref4 + (idx >> 3);
  uint32_t const sh = 4 * (idx & 7u);
  hint_compare(row, seq_stride, refw, sh, L, r, h);
  // ---- every k-mer: the label of its place, no label at all, or not provable
  HintKmer const k0 = hint_kmer<0>(ix, idx, seq4, r, h), k1 = hint_kmer<1>(ix, idx, seq4, r, h);
  HintKmer const none{HINT_K_HOLE, HINT_NO_SITE, 0u, false, false};
  HintKmer const k2 = n_k > 2 ? hint_kmer<2>(ix, idx, seq4, r, h) : none;
  HintKmer const k3 = n_k > 3 ? hint_kmer<3>(ix, idx, seq4, r, h) : none;
  HintKmer const k4 = n_k > 4 ? hint_kmer<4>(ix, idx, seq4, r, h) : none;
  if (k0.kind == HINT_K_DECLINE || k1.kind == HINT_K_DECLINE || k2.kind == HINT_K_DECLINE || k3.kind == HINT_K_DECLINE ||
      k4.kind == HINT_K_DECLINE)
    return false;
  uint32_t const labelled = (k0.kind == HINT_K_LABEL ? 1u : 0u) | (k1.kind == HINT_K_LABEL ? 2u : 0u) | (k2.kind == HINT_K_LABEL ? 4u : 0u) |
                            (k3.kind == HINT_K_LABEL ? 8u : 0u) | (k4.kind == HINT_K_LABEL ? 16u : 0u);
  uint32_t const par = (k0.par ? 1u : 0u) | (k1.par ? 2u : 0u) | (k2.par ? 4u : 0u) | (k3.par ? 8u : 0u) | (k4.par ? 16u : 0u);
  uint32_t const mmk = (k0.mm ? 1u : 0u) | (k1.mm ? 2u : 0u) | (k2.mm ? 4u : 0u) | (k3.mm ? 8u : 0u) | (k4.mm ? 16u : 0u);
  // ---- the run of k-mers that makes the path (express4.inl: the longest run of labelled k-mers, which has to be the
  //      only one of its length; the shorter side of a hole chains into a path remove_short_paths drops)
  uint32_t lo = 0, hi = n_k - 1;
  if (labelled != (1u << n_k) - 1u)
  {
    uint32_t best_lo = 0, best_len = 0, second = 0, cur_lo = 0, cur_len = 0;
#pragma unroll
    for (uint32_t k = 0; k <= AlignCfg::KC; ++k)
    {
      if (k < n_k && ((labelled >> k) & 1u))
      {
        cur_lo = cur_len == 0 ? k : cur_lo;
        ++cur_len;
      }
      else
      {
        if (cur_len > best_len)
        {
          second = best_len;
          best_len = cur_len;
          best_lo = cur_lo;
        }
        else if (cur_len > second)
          second = cur_len;
        cur_len = 0;
      }
    }
    // (a run that opens with a parallel chain behind a hole is returned twice by the reference: not here)
    if (best_len <= second || (best_lo > 0 && ((par >> best_lo) & 1u)))
    {
      GTX_HINT_NOTE(10);
      return false;
    }
    lo = best_lo;
    hi = best_lo + best_len - 1;
  }
  uint32_t const run = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
  uint32_t mism = static_cast<uint32_t>(__builtin_popcount(mmk & run));
  // ---- the read in front of the run and behind it: the walks' shortcut, both inside the reference node the path touches.
  //      Mismatches in [0, x) for the k-mer boundaries x = 31 j:
  uint32_t const prs = (K - 1) * lo, pre = (K - 1) * (hi + 1);
  uint32_t start = g.first_order + idx + prs, rs = prs;
  if (prs != 0) // walk_read_starts (genotype_paths.cpp:555-621)
  {
    uint32_t const y = ix.pos_flags[idx + prs].y;
    if ((y & 255u) == 0 || ((y >> HINT_BACK_SHIFT) & 255u) < prs)
    {
      GTX_HINT_NOTE(11);
      return false; // (the walk leaves the node: express4 / general pass)
    }
    uint32_t const head_len = prs + 1;
    // mismatches in [0, prs]: the boundary base itself is nibble 31 lo
    uint32_t const upto = lo == 1 ? h.upto[1] + h.edge[1] : lo == 2 ? h.upto[2] + h.edge[2] : lo == 3 ? h.upto[3] + h.edge[3] : h.upto[4] + h.edge[4];
    uint32_t const budget = 2 + head_len / 11 < 7 ? 2 + head_len / 11 : 7; // genotype_paths.cpp:571-577
    if (upto <= budget)
    {
      start -= prs;
      rs = 0;
      mism += upto;
    }
  }
  uint32_t end = g.first_order + idx + pre, re = pre;
  if (pre != L - 1) // walk_read_ends (genotype_paths.cpp:483-553)
  {
    uint32_t const tail_len = L - pre;
    uint32_t const room = ix.pos_flags[idx + pre].y & 255u;
    if (room < tail_len)
    {
      GTX_HINT_NOTE(8);
      return false; // (the tail leaves the node, or the path ends on a variant: express4)
    }
    uint32_t const before = hi == 0 ? h.upto[1] : hi == 1 ? h.upto[2] : hi == 2 ? h.upto[3] : hi == 3 ? h.upto[4] : h.upto[5];
    uint32_t const got = h.upto[0] - before;
    uint32_t const budget = 2 + tail_len / 11 < 7 ? 2 + tail_len / 11 : 7; // genotype_paths.cpp:505-511
    if (got <= budget)
    {
      re = L - 1;
      mism += got;
      end += tail_len - 1;
    }
  }
  // ---- variant sites of the path, most recent k-mer first (Path(p1, p2), path.cpp:38-82); a site under two
  //      neighbouring k-mers is one entry (the same base, hence the same allele)
  uint32_t vs[5], va[5];
  uint32_t nvar = 0;
  uint32_t last = HINT_NO_SITE;
  bool clash = false;
  auto push = [&](uint32_t k, HintKmer const & km)
  {
    if (((run >> k) & 1u) && km.site != HINT_NO_SITE)
    {
      if (km.site == last)
        clash = clash || va[nvar - 1] != km.allele;
      else
      {
        vs[nvar] = km.site;
        va[nvar] = km.allele;
        ++nvar;
        last = km.site;
      }
    }
  };
  push(4, k4);
  push(3, k3);
  push(2, k2);
  push(1, k1);
  push(0, k0);
  if (clash || 6 + 3 * nvar > rec_words)
    return false;
  uint32_t np = 1, longest = re - rs + 1;
  if (mism > 10) // remove_paths_with_too_many_mismatches
  {
    np = 0;
    longest = 0;
  }
  rec[0] = np;
  rec[1] = longest | (L << 16) | ((np && nvar) ? GTX_REC_HAS_VARIANTS : 0u);
  if (np)
  {
    rec[2] = start;
    rec[3] = end;
    rec[4] = rs | (re << 16);
    rec[5] = mism | (nvar << 16);
    for (uint32_t k = 0; k < 5; ++k)
      if (k < nvar)
      {
        rec[6 + 3 * k] = vs[k];
        rec[7 + 3 * k] = 1u << va[k];
        rec[8 + 3 * k] = 0u;
      }
  }
  return true;
}

} // namespace gtx
