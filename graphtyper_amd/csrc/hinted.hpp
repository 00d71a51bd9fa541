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
GTX_DEV uint32_t nib_range_mask(uint32_t w, uint32_t a, uint32_t b)
{
  uint32_t const lo = a > 8 * w ? a - 8 * w : 0u, hi = b > 8 * w ? b - 8 * w : 0u; // nibbles [lo, hi) of this word, clamped below
  uint32_t const from = lo >= 8 ? 0u : 0x11111111u >> (4 * lo); // nibbles lo..7
  uint32_t const upto = hi >= 8 ? 0u : 0x11111111u >> (4 * hi); // nibbles hi..7
  return from & ~upto;
}

template <uint32_t A, uint32_t B>
GTX_DEV uint32_t nib_count(uint32_t const (&x)[HINT_WORDS])
{
  uint32_t c = 0;
#pragma unroll
  for (uint32_t w = A / 8; w < (B + 7) / 8 && w < HINT_WORDS; ++w)
    c += static_cast<uint32_t>(__builtin_popcount(x[w] & nib_range_mask(w, A, B)));
  return c;
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

struct HintKmer // what the compare says about k-mer I of the read
{
  uint32_t mis, mis_left, amb, amb_left, amb_outside; // counts over its 32 bases (left = the 16 first)
};

template <uint32_t I>
GTX_DEV bool hint_kmer_ok(IndexView const & ix, uint32_t idx, uint32_t const (&r)[HINT_WORDS], uint32_t const (&mk)[HINT_WORDS],
                          uint32_t const (&am)[HINT_WORDS], uint32_t const (&out)[HINT_WORDS], bool & mm, uint32_t & site)
{
  constexpr uint32_t A = (K - 1) * I;
  uint32_t const f = ix.pos_flags[idx + A];
  uint32_t const mis = nib_count<A, A + 32>(mk), mis_left = nib_count<A, A + 16>(mk);
  uint32_t const amb = nib_count<A, A + 32>(am), amb_left = nib_count<A, A + 16>(am);
  uint32_t const amb_out = nib_count<A, A + 32>(out); // ambiguous bases whose set does not hold the reference base
  site = f >> HINT_SITE_SHIFT;
  mm = false;
  if (amb == 0 && mis == 0)
    return (f & HINT_EXACT_OK) != 0;
  if ((f & HINT_SINGLE_OK) == 0)
    return false;
  if (amb == 0 && mis == 1)
  {
    // the half with the substitution must occur in no indexed key, the other half in K only
    bool const left = mis_left == 1;
    if ((f & (left ? HINT_R1 : HINT_L1)) == 0)
      return false;
    uint32_t l0, l1, r0, r1;
    nib_extract16<A>(r, l0, l1);
    nib_extract16<A + 16>(r, r0, r1);
    uint32_t const bit = hint_filter_bit(left ? l0 : r0, left ? l1 : r1, ix.filt_log2);
    mm = true;
    return ((ix.filt[left ? 0 : 1][bit >> 5] >> (bit & 31u)) & 1u) == 0;
  }
  if (amb == 1 && mis == 0 && amb_out == 0)
    return (f & (amb_left == 1 ? HINT_R1 : HINT_L1)) != 0;
  return false;
}

// The forward task of one read.  Returns true when the record was written, false = declined (nothing written).
GTX_DEV bool hinted_one(GraphView const & g, IndexView const & ix, uint8_t const * seq4, uint32_t seq_stride, gtx_read_meta const & m,
                        uint32_t * rec, uint32_t rec_words)
{
  uint32_t const L = m.l_qseq;
  if (L < 2 * K - 1 || L > HINT_MAX_READ || m.pos < 0 || ix.n_hint == 0)
    return false;
  // position of read base 0 in the hint tables
  if (static_cast<uint32_t>(m.pos) < ix.hint_first)
    return false;
  uint32_t const idx = static_cast<uint32_t>(m.pos) - ix.hint_first;
  if (idx >= ix.n_hint || L > ix.n_hint - idx)
    return false;
  uint32_t const n_k = 1 + (L - K) / (K - 1), pre = (K - 1) * n_k;
  // ---- the read and the reference under it, 8 bases per word, aligned to the read
  uint32_t r[HINT_WORDS], mk[HINT_WORDS], am[HINT_WORDS], ao[HINT_WORDS], mt[HINT_WORDS];
  uint32_t const * seqw = reinterpret_cast<uint32_t const *>(seq4);
  uint32_t const * refw = ix.ref4 + (idx >> 3);
  uint32_t const sh = 4 * (idx & 7u);
  uint32_t prev = refw[0];
#pragma unroll
  for (uint32_t w = 0; w < HINT_WORDS; ++w)
  {
    uint32_t const have = L > 8 * w ? L - 8 * w : 0u; // bases of the read in this word
    uint32_t rw = 0, gw = 0;
    if (have != 0 && 4 * w < seq_stride)
      rw = hint_bswap(seqw[w]);
    uint32_t const next = have != 0 ? refw[w + 1] : 0u;
    if (have != 0)
      gw = sh == 0 ? prev : ((prev << sh) | (next >> (32 - sh)));
    prev = next;
    uint32_t const keep = have >= 8 ? 0xFFFFFFFFu : have == 0 ? 0u : ~(0xFFFFFFFFu >> (4 * have));
    rw &= keep;
    gw &= keep;
    uint32_t const flags = keep & 0x11111111u;
    uint32_t const differ = nib_nonzero(rw ^ gw);
    uint32_t const r_amb = nib_not_onehot(rw) & flags;                     // '=' (0), N and every other IUPAC set
    uint32_t const r_any = (nib_is15(rw) | (~nib_nonzero(rw) & flags));    // N or '=' (which the reference reads as N)
    r[w] = rw;
    am[w] = r_amb;
    mk[w] = differ & ~r_amb;
    ao[w] = r_amb & ~(nib_nonzero(rw & gw) | r_any);                       // the set misses the reference base
    mt[w] = differ & ~r_any & ~nib_is15(gw);                               // count_mismatches (graph_utils.hpp:7-69)
  }
  // ---- every k-mer settled by the flags of its place?
  bool mm0 = false, mm1 = false, mm2 = false, mm3 = false, mm4 = false;
  uint32_t s0 = HINT_NO_SITE, s1 = HINT_NO_SITE, s2 = HINT_NO_SITE, s3 = HINT_NO_SITE, s4 = HINT_NO_SITE;
  bool ok = hint_kmer_ok<0>(ix, idx, r, mk, am, ao, mm0, s0) && hint_kmer_ok<1>(ix, idx, r, mk, am, ao, mm1, s1);
  if (ok && n_k > 2)
    ok = hint_kmer_ok<2>(ix, idx, r, mk, am, ao, mm2, s2);
  if (ok && n_k > 3)
    ok = hint_kmer_ok<3>(ix, idx, r, mk, am, ao, mm3, s3);
  if (ok && n_k > 4)
    ok = hint_kmer_ok<4>(ix, idx, r, mk, am, ao, mm4, s4);
  if (!ok)
    return false;
  uint32_t mism = (mm0 ? 1u : 0u) + (mm1 ? 1u : 0u) + (n_k > 2 && mm2 ? 1u : 0u) + (n_k > 3 && mm3 ? 1u : 0u) + (n_k > 4 && mm4 ? 1u : 0u);
  // ---- the rest of the read behind the last k-mer (walk_read_ends through its shortcut: inside the reference node)
  uint32_t const start = g.first_order + idx;
  uint32_t end = start + pre, re = pre;
  if (pre != L - 1)
  {
    uint32_t const tail_len = L - pre;
    uint32_t const room = (ix.pos_flags[idx + pre] >> HINT_ROOM_SHIFT) & 255u;
    if (room < tail_len)
      return false; // (the tail leaves the node, or the path ends on a variant: express4)
    uint32_t const all = nib_count<0, HINT_MAX_READ>(mt);
    uint32_t const before = n_k == 2 ? nib_count<0, 2 * (K - 1)>(mt) : n_k == 3 ? nib_count<0, 3 * (K - 1)>(mt) :
                            n_k == 4 ? nib_count<0, 4 * (K - 1)>(mt) : nib_count<0, 5 * (K - 1)>(mt);
    uint32_t const got = all - before;
    uint32_t const budget = 2 + tail_len / 11 < 7 ? 2 + tail_len / 11 : 7; // genotype_paths.cpp:505-511
    if (got <= budget)
    {
      re = L - 1;
      mism += got;
      end += tail_len - 1;
    }
  }
  // ---- variant sites of the path, most recent k-mer first (Path(p1, p2), path.cpp:38-82); every label here names the
  //      reference allele, a site under two neighbouring k-mers is one entry
  uint32_t vs[5];
  uint32_t nvar = 0;
  uint32_t last = HINT_NO_SITE;
  auto push = [&](bool on, uint32_t s)
  {
    if (on && s != HINT_NO_SITE && s != last)
    {
      vs[nvar++] = s;
      last = s;
    }
  };
  push(n_k > 4, s4);
  push(n_k > 3, s3);
  push(n_k > 2, s2);
  push(true, s1);
  push(true, s0);
  if (6 + 3 * nvar > rec_words)
    return false;
  uint32_t np = 1, longest = re + 1;
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
    rec[4] = re << 16; // read_start_index 0
    rec[5] = mism | (nvar << 16);
    for (uint32_t k = 0; k < 5; ++k)
      if (k < nvar)
      {
        rec[6 + 3 * k] = vs[k];
        rec[7 + 3 * k] = 1u; // allele 0
        rec[8 + 3 * k] = 0u;
      }
  }
  return true;
}

} // namespace gtx
