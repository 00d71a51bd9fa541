// hinted.hpp -- pass 0 of the alignment: ONE READ PER LANE, guided by the position the read comes with.
//
// The reference looks every k-mer of a read up globally -- the exact key and its 96 Hamming-1 neighbours
// (src/utilities/kmer_help_functions.cpp:53-119, src/index/ph_index.cpp:66-107) -- because it does not use where the
// mapper put the read.  A record of a sorted BAM does carry that place (bam1_t::core.pos, handed over as
// gtx_read_meta::pos).  This pass compares the read with the linear reference AT that place, bit-parallel on packed
// planes, and takes the answers of the global lookups from flags computed once per reference position when the index
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
#include <type_traits>

#include "graph_dev.hpp"

namespace gtx
{
#ifdef GTX_EMU_NOTES // diagnostics of the host emulation (tests/emu): why a read leaves this pass
void hint_note(uint32_t code);
#define GTX_HINT_NOTE(code) hint_note(code)
#else
#define GTX_HINT_NOTE(code) ((void)0)
#endif

// (experiment builds only, -DGTX_X_STOP_AT=n: hinted_on_path ends behind its n-th part with a value that hangs on what the part
//  computed -- wrong results by design; the kernel's instruction counters then say what the parts up to there cost.  tools/pmc_main.sh)
#ifdef GTX_X_STOP_AT
#define GTX_X_STOP(n, v)                                                                                                          \
  if (GTX_X_STOP_AT == (n))                                                                                                      \
  return static_cast<uint32_t>((v) == 0xA5A5A5A5u ? 2u : 0u)
#else
#define GTX_X_STOP(n, v) ((void)0)
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

// the 2-bit codes of 16 unambiguous bases of the read from base A on (A is a compile-time constant inside a k-mer, hence
// inside the read), as the two planes hint_filter_slot hashes: lo / hi = low / high bit of A0 C1 G2 T3, base A + j at bit j.
// `row` is the read in plane form (graph_dev.hpp): C and T have bit 0 of the code pair set (planes 1, 3), G and T bit 1 (2, 3).
template <uint32_t A, class Row>
GTX_DEV void plane_extract16(Row row, uint32_t & lo, uint32_t & hi)
{
  constexpr uint32_t W = A / 32, S = A % 32;
  auto ext = [&](uint32_t b) -> uint32_t
  {
    uint32_t const a = row[4 * W + b];
    if constexpr (S <= 16)
      return (a >> S) & 0xFFFFu;
    else
      return ((a >> S) | (row[4 * (W + 1) + b] << (32 - S))) & 0xFFFFu;
  };
  uint32_t const p1 = ext(1), p2 = ext(2), p3 = ext(3);
  lo = p1 | p3;
  hi = p2 | p3;
}

// ... from a base A known at run time only (A + 15 inside the row's groups of 32 bases; `groups`: how many the row has)
template <class Row>
GTX_DEV void plane_extract16_at(Row row, uint32_t A, uint32_t & lo, uint32_t & hi, uint32_t groups = HINT_MAX_READ / 32)
{
  uint32_t const W = A >> 5, S = A & 31u, N = W + 1u < groups ? W + 1u : W; // (the last group is looked at with S <= 16 only)
  uint32_t const a1 = row[4 * W + 1], a2 = row[4 * W + 2], a3 = row[4 * W + 3], n1 = row[4 * N + 1], n2 = row[4 * N + 2], n3 = row[4 * N + 3];
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t const p1 = __builtin_amdgcn_alignbit(n1, a1, S), p2 = __builtin_amdgcn_alignbit(n2, a2, S), p3 = __builtin_amdgcn_alignbit(n3, a3, S);
#else
  uint32_t const p1 = S == 0 ? a1 : (a1 >> S) | (n1 << (32 - S)), p2 = S == 0 ? a2 : (a2 >> S) | (n2 << (32 - S)), p3 = S == 0 ? a3 : (a3 >> S) | (n3 << (32 - S));
#endif
  lo = (p1 | p3) & 0xFFFFu;
  hi = (p2 | p3) & 0xFFFFu;
}

// What the compare of the read with the reference under it says, summed while the words stream by.  Packed (the pass is
// bound by memory latency times resident waves: registers are occupancy):
//   k[I]   per k-mer, 6 bits each: substitutions among its unambiguous bases, ... in its 16 first bases, ambiguous bases,
//          ... in its 16 first bases, ambiguous bases whose set does not hold the reference base
//   upto   mismatches by the walks' rule in [0, 31 j), 8 bits each for j = 1..4;  more: j = 5 (bits 0..7), the whole
//          read (bits 8..15), and the mismatch flags of the boundary bases 31 j, j = 1..4 (bits 16..19)
struct HintCounts
{
  uint32_t k[AlignCfg::KC];
  uint32_t upto, more;
};
constexpr uint32_t HC_MIS = 0, HC_MIS_LEFT = 6, HC_AMB = 12, HC_AMB_LEFT = 18, HC_AMB_OUT = 24;

GTX_DEV uint32_t hc_get(uint32_t packed, uint32_t shift)
{
  return (packed >> shift) & 63u;
}

template <uint32_t W, uint32_t I>
GTX_DEV void hint_count_kmer(uint32_t mk, uint32_t am, uint32_t ao, uint32_t mt, HintCounts & h)
{
  constexpr uint32_t A = (K - 1) * I;
  constexpr uint32_t M = nib_range_mask(W, A, A + 32), ML = nib_range_mask(W, A, A + 16);
  constexpr uint32_t C = nib_range_mask(W, 0, A + (K - 1)), E = I == 0 ? 0u : nib_range_mask(W, A, A + 1);
  if constexpr (M != 0)
    h.k[I] += static_cast<uint32_t>(__builtin_popcount(mk & M)) + (static_cast<uint32_t>(__builtin_popcount(am & M)) << HC_AMB) +
              (static_cast<uint32_t>(__builtin_popcount(ao & M)) << HC_AMB_OUT);
  if constexpr (ML != 0)
    h.k[I] += (static_cast<uint32_t>(__builtin_popcount(mk & ML)) << HC_MIS_LEFT) + (static_cast<uint32_t>(__builtin_popcount(am & ML)) << HC_AMB_LEFT);
  if constexpr (C != 0)
  {
    if constexpr (I < 4)
      h.upto += static_cast<uint32_t>(__builtin_popcount(mt & C)) << (8 * I);
    else
      h.more += static_cast<uint32_t>(__builtin_popcount(mt & C));
  }
  if constexpr (E != 0)
    h.more += static_cast<uint32_t>(__builtin_popcount(mt & E)) << (15 + I);
}

// mismatches in [0, 31 j), j = 1..5 / in the whole read / at base 31 j, j = 1..4
GTX_DEV uint32_t hc_upto(HintCounts const & h, uint32_t j)
{
  return j == 5 ? (h.more & 255u) : ((h.upto >> (8 * (j - 1))) & 255u);
}

GTX_DEV uint32_t hc_all(HintCounts const & h)
{
  return (h.more >> 8) & 255u;
}

GTX_DEV uint32_t hc_edge(HintCounts const & h, uint32_t j)
{
  return (h.more >> (15 + j)) & 1u;
}

// word W of the read (rw: 8 bases, base j in bits 28-4j; `have` bases of the read lie in it) against the reference (gw)
template <uint32_t W>
GTX_DEV void hint_word(uint32_t rw, uint32_t gw, uint32_t have, HintCounts & h)
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
  h.more += static_cast<uint32_t>(__builtin_popcount(mt)) << 8;
}

// Pins the counters in registers at this point.  Without it the optimiser sinks the per-word arithmetic to where the
// counters are read -- behind all the loads' uses -- and keeps the forty loaded words alive until then: 130 VGPRs and 3
// waves per SIMD.  (The pass is bound by memory latency times resident waves: registers are occupancy.)
#if defined(__HIP_DEVICE_COMPILE__)
#define GTX_PIN(x) asm volatile("" : "+v"(x))
#else
#define GTX_PIN(x) ((void)0)
#endif

GTX_DEV void hint_pin(HintCounts & h)
{
  GTX_PIN(h.k[0]);
  GTX_PIN(h.k[1]);
  GTX_PIN(h.k[2]);
  GTX_PIN(h.k[3]);
  GTX_PIN(h.k[4]);
  GTX_PIN(h.upto);
  GTX_PIN(h.more);
}

template <uint32_t W>
GTX_DEV void hint_compare_words(uint32_t const (&rr)[HINT_WORDS], uint32_t const (&gg)[HINT_WORDS + 1], uint32_t sh, uint32_t L, HintCounts & h)
{
  if constexpr (W < HINT_WORDS)
  {
    uint32_t const have = L > 8 * W ? L - 8 * W : 0u; // bases of the read in this word
    uint32_t const gw = sh == 0 ? gg[W] : ((gg[W] << sh) | (gg[W + 1] >> (32 - sh)));
    hint_word<W>(hint_bswap(rr[W]), gw, have, h);
    hint_pin(h);
    hint_compare_words<W + 1>(rr, gg, sh, L, h);
  }
}

// The compare on nibble words (8 bases per word, read and reference alike): the first form of this pass, ~1 300 vector
// instructions per read.  Kept as the definition the plane form below is tested against (tests/test_hint_compare.py).
// refw: reference nibble words from the word that holds the read's first base (base 8w+j in bits 28-4j), sh = 4 * (phase).
template <class Row>
GTX_DEV void hint_compare_nibbles(Row row, uint32_t seq_stride, uint32_t const * refw, uint32_t sh, uint32_t L, HintCounts & h)
{
  uint32_t rr[HINT_WORDS], gg[HINT_WORDS + 1];
#pragma unroll
  for (uint32_t w = 0; w <= HINT_WORDS; ++w)
    gg[w] = refw[w];
#pragma unroll
  for (uint32_t w = 0; w < HINT_WORDS; ++w)
  {
    bool const in_row = 4 * w < seq_stride; // (uniform; a select, not a branch around the load)
    rr[w] = row[in_row ? w : 0u] & (in_row ? 0xFFFFFFFFu : 0u);
  }
  hint_compare_words<0>(rr, gg, sh, L, h);
}

// ---- the compare on bit planes.  Per-nibble questions (differs? one base or a set? N?) cost a handful of shifts and
// masks per 8 bases on nibble words; on four 1-bit planes (bit b of every base's code) the same questions are plain
// bitwise operations over 32 bases at once.  The reference is stored as planes when the index is built (IndexView::refp);
// the read's nibbles are transposed here: the four bits `b` of the nibbles of a 16-bit half are gathered into one nibble
// by ONE 24-bit multiply (the partial products of the four source bits land on distinct bits, so nothing carries), two
// halves make a byte (8 bases), four words a plane word (32 bases).
constexpr uint32_t HINT_PLANE_WORDS = HINT_MAX_READ / 32;
static_assert(HINT_MAX_READ % 32 == 0 && HINT_PLANE_WORDS == AlignCfg::KC, "five k-mers, five plane words");

GTX_DEV uint32_t hint_funnel(uint32_t lo, uint32_t hi, uint32_t s) // bits [s, s + 32) of hi:lo, s in 0..31
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return s == 0 ? lo : (lo >> s) | (hi << (32 - s));
#endif
}

// refp: the reference planes from the group of 32 positions that holds the read's first base (4 words per group: planes
// 0..3, position 32q+j at bit j), s = that base's bit in its group.  row: the read in the same form (graph_dev.hpp), a row of
// seq_stride bytes.  Same counters as hint_compare_nibbles.
template <class Row>
GTX_DEV void hint_compare(Row row, uint32_t seq_stride, uint32_t const * refp, uint32_t s, uint32_t L, HintCounts & h)
{
  uint32_t mk[HINT_PLANE_WORDS], am[HINT_PLANE_WORDS], ao[HINT_PLANE_WORDS], mt[HINT_PLANE_WORDS + 1];
  // all loads first (no branches around them: the plane array is padded, a row has at least seq_stride bytes)
  // (the read's words come from the row -- LDS in the kernel -- group by group inside the loop: only the reference words,
  //  a global round trip, are worth holding all at once)
  uint32_t gg[4 * (HINT_PLANE_WORDS + 1)];
#pragma unroll
  for (uint32_t w = 0; w < 4 * (HINT_PLANE_WORDS + 1); ++w)
    gg[w] = refp[w];
#pragma unroll
  for (uint32_t W = 0; W < HINT_PLANE_WORDS; ++W)
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
  mt[HINT_PLANE_WORDS] = 0;
  // ---- counters.  k-mer I is bases [31 I, 31 I + 32): a 32-bit window of the flag words (I = 0: word 0)
  uint32_t prefix = 0; // mismatches in the plane words in front of the current one
#pragma unroll
  for (uint32_t I = 0; I < AlignCfg::KC; ++I)
  {
    uint32_t const W0 = ((K - 1) * I) / 32, o = ((K - 1) * I) % 32;
    uint32_t const nxt = W0 + 1 < HINT_PLANE_WORDS ? W0 + 1 : W0; // (o = 0 only for I = 0: the next word is not looked at)
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
    else
      h.more |= upto;
    prefix += static_cast<uint32_t>(__builtin_popcount(mt[I]));
    if (I > 0) // the boundary base 31 I
      h.more |= ((mt[W0] >> o) & 1u) << (15 + I);
  }
  h.more |= prefix << 8; // the whole read
}

enum : uint32_t
{
  HINT_K_DECLINE = 0,
  HINT_K_LABEL = 1, // the k-mer's lists hold exactly the label of this place
  HINT_K_HOLE = 2   // the k-mer has no label at all
};

// one k-mer's verdict in a word: kind (bits 0..1), mm (2: the label comes from the Hamming-1 list, one more mismatch), par (3:
// the k-mer also starts a parallel chain -- matters when it opens the run behind a hole), allele (4..5) and site (16..31,
// HINT_NO_SITE: none) of the label; bits 8..15: the allele set of a k-mer with several labels (HINT_MULTI), else 0
constexpr uint32_t HK_MM = 4u, HK_PAR = 8u, HK_ALLELE_SHIFT = 4u, HK_SET_SHIFT = 8u, HK_SITE_SHIFT = 16u;
// HK_TWO (dense build; in the allele field, which a verdict with sets does not use): the labels lie on the sites `site` and
// `site + 1`, bits 8..11 / 12..15 are the sets of their alleles (HINT_TWO)
constexpr uint32_t HK_TWO = 1u << HK_ALLELE_SHIFT;
GTX_DEV bool hk_is_two(uint32_t km) // (a verdict that names one allele has no set)
{
  return (km & HK_TWO) != 0 && ((km >> HK_SET_SHIFT) & 255u) != 0;
}

GTX_DEV uint32_t hk_make(uint32_t kind, uint32_t site, uint32_t allele, bool mm, bool par)
{
  return kind | (mm ? HK_MM : 0u) | (par ? HK_PAR : 0u) | (allele << HK_ALLELE_SHIFT) | (site << HK_SITE_SHIFT);
}

// What the global lookups of the reference return for k-mer I of the read, proven from the flags of the hinted place
// (cases in the file header; a hole needs both halves of the k-mer to occur in no indexed key, or to be K's own halves
// while the k-mer is two or more substitutions away from K).
// What the global lookups of the reference return for k-mer I of the read, as far as the flags of the hinted place
// say.  The verdict may hang on one or both halves of the k-mer occurring in no indexed key (HK_NEED_LEFT / RIGHT): the
// caller probes the filters for all k-mers at once -- one memory round trip instead of one per k-mer -- and turns the
// verdict into a decline when a needed half may occur.
constexpr uint32_t HK_NEED_LEFT = 64u, HK_NEED_RIGHT = 128u;

template <uint32_t I, bool DENSE, class Row, class Counts>
GTX_DEV uint32_t hint_kmer_judge(uint2_t const f, Row row, Counts const & h, uint32_t & amb2) // (Counts: HintCounts, or hinted_long.hpp's for eight k-mers)
{
  constexpr uint32_t A = (K - 1) * I;
  uint32_t mis = hc_get(h.k[I], HC_MIS), mis_left = hc_get(h.k[I], HC_MIS_LEFT);
  uint32_t amb = hc_get(h.k[I], HC_AMB), amb_left = hc_get(h.k[I], HC_AMB_LEFT);
  uint32_t amb_out = hc_get(h.k[I], HC_AMB_OUT); // ambiguous bases whose set does not hold the reference base
  uint32_t const site = f.x >> HINT_SITE_SHIFT;
  uint32_t const declined = hk_make(HINT_K_DECLINE, site, 0u, false, false);
  // K's own label list is known: the one label of its place (HINT_SINGLE_OK), or -- dense build -- the labels of several
  // alleles of one site / of two sites on K's interval (HINT_MULTI, HINT_TWO: `own_set` goes into every verdict that names them)
  bool const several = DENSE && (f.x & (HINT_MULTI | HINT_TWO)) != 0 && (f.x & HINT_EXACT_OK) != 0;
  bool const single = (f.x & HINT_SINGLE_OK) != 0 || several;
  uint32_t const own_set = several ? (((f.x >> HINT_ALTIDX_SHIFT) & 255u) << HK_SET_SHIFT) | ((f.x & HINT_TWO) ? HK_TWO : 0u) : 0u;
  // gl / gr: who else has the 16 first / last bases of the key the read's k-mer is judged against is known -- that key
  // alone, or (on the far side of a SNP under the k-mer, HINT_SNP_GROUP) the keys of the SNP's alleles and nobody else
  bool gl = single && (f.x & HINT_L1) != 0, gr = single && (f.x & HINT_R1) != 0;
  // ... or (dense build) the others are too far from K to be met by a k-mer m substitutions from it (HINT_FAR_*)
  uint32_t const far_l = DENSE && single ? (f.y >> HINT_FAR_LEFT_SHIFT) & 3u : 0u, far_r = DENSE && single ? (f.y >> HINT_FAR_RIGHT_SHIFT) & 3u : 0u;
  auto GL = [&](uint32_t m) { return gl || far_l >= hint_far_need(m); };
  auto GR = [&](uint32_t m) { return gr || far_r >= hint_far_need(m); };
  if (amb == 0 && mis == 0)
  {
    if constexpr (DENSE)
      if ((f.x & HINT_TWO) != 0) // (two sites under the k-mer)
        return hk_make(HINT_K_LABEL, site, 0u, false, (f.x & HINT_PAR) != 0) | HK_TWO | (((f.x >> HINT_ALTIDX_SHIFT) & 255u) << HK_SET_SHIFT);
    // (lean build: a place with HINT_TWO is like one without HINT_EXACT_OK)
    bool const exact_ok = DENSE ? (f.x & HINT_EXACT_OK) != 0 : (f.x & (HINT_EXACT_OK | HINT_TWO)) == HINT_EXACT_OK;
    GTX_HINT_NOTE(exact_ok ? 0 : 1); // exact k-mer, but the place is not provably simple
    uint32_t const set = (f.x & HINT_MULTI) ? ((f.x >> HINT_ALTIDX_SHIFT) & 255u) << HK_SET_SHIFT : 0u; // (several alleles of a merged site)
    return hk_make(exact_ok ? HINT_K_LABEL : HINT_K_DECLINE, site, 0u, false, (f.x & HINT_PAR) != 0) | set;
  }
  // ---- a SNP under the k-mer: the read is judged against the key of the allele it carries.  The compare above ran against
  //      the reference allele: a base on the site that is another allele's comes off the counters again.
  uint32_t allele = 0, site_set = 0;
  bool third = false, site_amb = false, site_left = false;
  if ((f.x & HINT_ALT_OK) != 0)
  {
    uint32_t const off = (f.y >> HINT_SNPOFF_SHIFT) & 31u, at = A + off;
    uint32_t const rb0 = plane_code_at(row, at), rb = rb0 == 0 ? 15u : rb0; // ('=' reads as N)
    bool const onehot = (rb & (rb - 1u)) == 0;
    uint32_t const two = rb == 1 ? 0u : rb == 2 ? 1u : rb == 4 ? 2u : 3u, refb = (f.y >> HINT_REFB_SHIFT) & 3u;
    bool const group = (f.y & HINT_SNP_GROUP) != 0;
    if (onehot && two != refb)
    {
      uint32_t const a = (f.x >> (HINT_ALTIDX_SHIFT + 2 * two)) & 3u;
      if (amb == 0 && mis == 1 && a != 0)
        return hk_make(HINT_K_LABEL, site, a, false, true); // exactly the other allele's key (par: the reference allele's key is its neighbour)
      if (group)
      {
        --mis;
        mis_left -= off < K / 2 ? 1u : 0u;
        allele = a;
        third = a == 0; // a base no allele has: one difference against every allele's key
      }
    }
    else if (!onehot && group)
    {
      // an ambiguous base ON the site: its expansion runs over the alleles whose base the set holds
      site_amb = true;
      --amb;
      amb_left -= off < K / 2 ? 1u : 0u;
      amb_out -= ((rb >> refb) & 1u) ? 0u : 1u;
      site_set = (rb >> refb) & 1u;
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b)
      {
        uint32_t const a = (f.x >> (HINT_ALTIDX_SHIFT + 2 * b)) & 3u;
        site_set |= (a != 0 && ((rb >> b) & 1u)) ? 1u << a : 0u;
      }
    }
    if (group)
    {
      gl = gr = true;
      site_left = off < K / 2;
    }
  }
  uint32_t const mis_right = mis - mis_left, amb_right = amb - amb_left;
  if (site_amb)
  {
    // a multi-key list (exact lookups only, src/utilities/kmer_help_functions.cpp:97-119).  Its keys run over the bases of
    // the set on the site; the half WITHOUT the site is shared with the allele keys only, so while that half is the
    // reference's, a key of the list is indexed iff it is an allele's key
    uint32_t const amb_far = site_left ? amb_right : amb_left, mis_far = site_left ? mis_right : mis_left;
    if (amb_far != 0 || amb > 1)
    {
      GTX_HINT_NOTE(7);
      return declined;
    }
    if (mis == 0)
      return hk_make((site_set != 0 && amb_out == 0) ? HINT_K_LABEL : HINT_K_HOLE, site, 0u, false, true) | (site_set << HK_SET_SHIFT);
    // substitutions: none of the keys is an allele's.  Far half clean: nobody else has it; else it is one concrete 16-mer to probe
    return hk_make(HINT_K_HOLE, site, 0u, false, true) | (mis_far == 0 ? 0u : site_left ? HK_NEED_RIGHT : HK_NEED_LEFT);
  }
  if (third)
  {
    uint32_t const need_site = site_left ? HK_NEED_LEFT : HK_NEED_RIGHT, need_far = site_left ? HK_NEED_RIGHT : HK_NEED_LEFT;
    uint32_t const amb_far = site_left ? amb_right : amb_left, mis_far = site_left ? mis_right : mis_left;
    if (amb == 0)
    {
      // one key.  No exact hit (the far half belongs to the allele keys, and this is none of them); its Hamming-1 neighbours:
      // every allele's key when nothing else differs -- their labels share the interval: one path with all alleles -- and
      // whoever else has the 16 bases around the foreign base: the filter has to say nobody
      if (mis == 0)
        return hk_make(HINT_K_LABEL, site, 0u, true, false) | (((1u << ((f.y >> HINT_NV_SHIFT) & 7u)) - 1u) << HK_SET_SHIFT) | need_site;
      return hk_make(HINT_K_HOLE, site, 0u, false, false) | need_site | (mis_far != 0 ? need_far : 0u);
    }
    // ambiguous bases besides: a multi-key list none of whose keys is an allele's
    if (amb_far == 0 && mis_far == 0)
      return hk_make(HINT_K_HOLE, site, 0u, false, true);
    if (amb - amb_far == 0)
      return hk_make(HINT_K_HOLE, site, 0u, false, true) | need_site;
    GTX_HINT_NOTE(12);
    return declined;
  }
  if (amb == 0 && mis == 0) // (the other allele's key with ambiguity taken off -- cannot happen -- or with the site's base only: handled above)
    return hk_make(HINT_K_LABEL, site, allele, false, true);
  if (amb == 2 && mis == 0 && (amb_left == 0 || amb_left == 2))
  {
    // two ambiguous bases in one half, the rest == K: the (up to 16) keys of the expansion all carry K's other half,
    // which K alone has -> of the expansion only K can be indexed, and it is in there when both sets hold its base
    // (... or whoever else has that half is three or more substitutions from K: the expansion's keys are within two)
    if (!(amb_left == 2 ? GR(1) : GL(1)))
    {
      GTX_HINT_NOTE(5);
      return declined;
    }
    return hk_make(amb_out == 0 ? HINT_K_LABEL : HINT_K_HOLE, site, allele, false, true) | own_set;
  }
  if (amb == 2 && mis == 0 && amb_left == 1 && GL(1))
  {
    // one ambiguous base in each half, the rest == K: of the (up to 16) keys those with K's base on the left carry K's left
    // half -- nobody else has it: K, or nothing --, the others carry one of three left halves a substitution away from K's,
    // and the caller asks the filter about those three (amb2: this k-mer): all absent -> of the list only K can be indexed
    amb2 |= 1u << I;
    return hk_make(amb_out == 0 ? HINT_K_LABEL : HINT_K_HOLE, site, allele, false, true) | own_set;
  }
  // Ambiguous bases AND substitutions: a multi-key list (exact lookups only, kmer_help_functions.cpp:97-119) every key of which
  // carries the half that holds no ambiguous base unchanged.  When that half has a substitution it is one concrete 16-mer, and if
  // it occurs in no indexed key (filter probe by the caller, or HINT_NEAR_FREE) none of the list's keys is indexed, whatever the
  // other half holds -- any number of ambiguous bases, further substitutions: the k-mer has no label.  (Round 6: with an N and a
  // substitution in one half and a substitution in the other the k-mer was sent on -- a third of what this pass declined at cfg2.)
  if (amb != 0 && mis != 0)
  {
    if (mis_left != 0 && amb_left == 0)
      return hk_make(HINT_K_HOLE, site, 0u, false, true) | HK_NEED_LEFT;
    if (mis_right != 0 && amb_right == 0)
      return hk_make(HINT_K_HOLE, site, 0u, false, true) | HK_NEED_RIGHT;
  }
  if (amb > 1)
  {
    GTX_HINT_NOTE(7);
    return declined;
  }
  // the k-mer is not K: a half without a difference is K's own -- nobody but K (or the SNP's allele keys) may have it
  // (flag), or nobody near enough to K (HINT_FAR_*) --, a half with one must occur in no indexed key (filter probe by the caller)
  if (amb == 0)
  {
    if (mis == 1)
    {
      bool const left = mis_left == 1;
      if (!(left ? GR(1) : GL(1)))
      {
        GTX_HINT_NOTE(3); // one substitution, the other half is shared with further keys (a variant there)
        return declined;
      }
      return hk_make(HINT_K_LABEL, site, allele, true, false) | (left ? HK_NEED_LEFT : HK_NEED_RIGHT) | own_set;
    }
    // two or more substitutions: no label at all (K itself is too far away to be a neighbour)
    bool const ok = (mis_left != 0 || GL(mis)) && (mis_right != 0 || GR(mis));
    GTX_HINT_NOTE(ok ? 0 : 6);
    return hk_make(ok ? HINT_K_HOLE : HINT_K_DECLINE, site, 0u, false, false) | (mis_left != 0 ? HK_NEED_LEFT : 0u) |
           (mis_right != 0 ? HK_NEED_RIGHT : 0u);
  }
  // one ambiguous base (a multi-key list: exact lookups only, src/utilities/kmer_help_functions.cpp:97-119)
  bool const amb_is_left = amb_left == 1;
  if (mis == 0)
  {
    // its keys differ from K in that base only: they share the other half with K
    if (!(amb_is_left ? GR(1) : GL(1)))
    {
      GTX_HINT_NOTE(5);
      return declined;
    }
    return hk_make(amb_out == 0 ? HINT_K_LABEL : HINT_K_HOLE, site, allele, false, true) | own_set; // (a set without the reference base: none of its keys is K)
  }
  // ... plus substitutions: none of its keys is K.  Either everything lies in one half (the other one is K's), or the
  // substitutions lie in the half without the ambiguous base, which then is one concrete 16-mer to probe
  uint32_t need = 0;
  bool ok = false;
  if (amb_is_left)
  {
    ok = mis_right == 0 ? GR(mis + 1u) : mis_left == 0;
    need = mis_right == 0 ? 0u : HK_NEED_RIGHT;
  }
  else
  {
    ok = mis_left == 0 ? GL(mis + 1u) : mis_right == 0;
    need = mis_left == 0 ? 0u : HK_NEED_LEFT;
  }
  GTX_HINT_NOTE(ok ? 0 : 12);
  return hk_make(ok ? HINT_K_HOLE : HINT_K_DECLINE, site, 0u, false, true) | (ok ? need : 0u);
}

// ... and the probes whose answer the tables already hold (HINT_NEAR_FREE, gtx_flat.hpp): a half of the read's k-mer that differs
// from the reference's in exactly one unambiguous base and holds no ambiguous one IS one of the 48 neighbours of that half the
// filter was asked about when the flags were made -- absent, every one of them.  The verdict no longer hangs on that half.
template <uint32_t I, bool DENSE, class Row, class Counts>
GTX_DEV uint32_t hint_kmer(uint2_t const f, Row row, Counts const & h, uint32_t & amb2)
{
  uint32_t v = hint_kmer_judge<I, DENSE>(f, row, h, amb2);
#ifndef GTX_NO_NEAR_FREE // (A/B build: every probe is made)
  if ((v & (HK_NEED_LEFT | HK_NEED_RIGHT)) != 0 && (f.y & HINT_NEAR_FREE) != 0)
  {
    uint32_t const mis_left = hc_get(h.k[I], HC_MIS_LEFT), mis_right = hc_get(h.k[I], HC_MIS) - mis_left;
    uint32_t const amb_left = hc_get(h.k[I], HC_AMB_LEFT), amb_right = hc_get(h.k[I], HC_AMB) - amb_left;
    if (mis_left == 1 && amb_left == 0)
      v &= ~HK_NEED_LEFT;
    if (mis_right == 1 && amb_right == 0)
      v &= ~HK_NEED_RIGHT;
  }
#endif
  return v;
}

// filter probe of one half of k-mer I when the verdict hangs on it: where to look ...
template <uint32_t I, uint32_t SIDE, class Row>
GTX_DEV void hint_probe_slot(IndexView const & ix, uint32_t verdict, Row row, uint32_t & word, uint32_t & mask)
{
  word = 0;
  mask = 0;
  if (verdict & (SIDE == 0 ? HK_NEED_LEFT : HK_NEED_RIGHT))
  {
    uint32_t w0, w1;
    plane_extract16<(K - 1) * I + 16 * SIDE>(row, w0, w1);
    hint_filter_slot(w0, w1, ix.filt_log2, word, mask);
  }
}

// ... and what the looked-up filter words say: the verdict stands, or the k-mer is declined
GTX_DEV uint32_t hint_probe_verdict(uint32_t verdict, uint32_t left_word, uint32_t left_mask, uint32_t right_word, uint32_t right_mask)
{
  bool const maybe = ((verdict & HK_NEED_LEFT) && (left_word & left_mask) == left_mask) ||
                     ((verdict & HK_NEED_RIGHT) && (right_word & right_mask) == right_mask);
  GTX_HINT_NOTE(maybe ? 4 : 0); // a half that has to be absent may occur in the index (a variant allele, or a filter collision)
  return maybe ? (verdict & ~3u) | HINT_K_DECLINE : verdict;
}

// ---- dense build: the walk at the read's end over one or two sites with alleles of any length ----
// mismatches by the walks' rule (count_mismatches, graph_utils.hpp:7-69) of read bases [ro, ro + n) against the linear
// reference from hint position q on (bit planes on both sides, 32 bases a round)
template <class Row>
GTX_DEV uint32_t hint_mm_linear(Row row, uint32_t const * refp, uint32_t ro, uint32_t q, uint32_t n)
{
  uint32_t mm = 0;
  for (uint32_t done = 0; done < n; done += 32)
  {
    uint32_t const a = ro + done, b = q + done, left = n - done;
    uint32_t const rw = a >> 5, rs = a & 31u, gw = b >> 5, gs = b & 31u;
    uint32_t const rw1 = rw + 1 < HINT_PLANE_WORDS ? rw + 1 : rw; // (what comes from it lies behind the read's row: masked by n)
    uint32_t const v = left >= 32 ? 0xFFFFFFFFu : (1u << left) - 1u;
    uint32_t const r0 = hint_funnel(row[4 * rw + 0], row[4 * rw1 + 0], rs), r1 = hint_funnel(row[4 * rw + 1], row[4 * rw1 + 1], rs);
    uint32_t const r2 = hint_funnel(row[4 * rw + 2], row[4 * rw1 + 2], rs), r3 = hint_funnel(row[4 * rw + 3], row[4 * rw1 + 3], rs);
    uint32_t const g0 = hint_funnel(refp[4 * gw + 0], refp[4 * gw + 4], gs), g1 = hint_funnel(refp[4 * gw + 1], refp[4 * gw + 5], gs);
    uint32_t const g2 = hint_funnel(refp[4 * gw + 2], refp[4 * gw + 6], gs), g3 = hint_funnel(refp[4 * gw + 3], refp[4 * gw + 7], gs);
    uint32_t const differ = (r0 ^ g0) | (r1 ^ g1) | (r2 ^ g2) | (r3 ^ g3);
    uint32_t const r_any = (r0 & r1 & r2 & r3) | ~(r0 | r1 | r2 | r3); // N or '=' (which the reference reads as N)
    mm += static_cast<uint32_t>(__builtin_popcount(differ & ~r_any & ~(g0 & g1 & g2 & g3) & v));
  }
  return mm;
}

// ... and against n bases of an allele (graph codes at `dna_off`); kill: a character that ends every walk (graph_dev.hpp)
template <class Row>
GTX_DEV uint32_t hint_mm_allele(GraphView const & g, Row row, uint32_t ro, uint32_t dna_off, uint32_t n, bool & kill)
{
  uint32_t mm = 0;
  for (uint32_t j = 0; j < n; ++j)
  {
    uint8_t const gc = reinterpret_cast<uint8_t const *>(g.dna)[dna_off + j];
    uint32_t const w = (ro + j) >> 5, sft = (ro + j) & 31u;
    uint32_t const rc0 = ((row[4 * w] >> sft) & 1u) | (((row[4 * w + 1] >> sft) & 1u) << 1) | (((row[4 * w + 2] >> sft) & 1u) << 2) |
                         (((row[4 * w + 3] >> sft) & 1u) << 3);
    uint32_t const rc = rc0 == 0 ? 15u : rc0;
    kill = kill || gc == DNA_KILL;
    mm += (gc != rc && rc != 15u && gc != 15u) ? 1u : 0u;
  }
  return mm;
}

// Graph::get_labels_forward (graph.cpp:1187-1439) from a position inside reference node r when the `T` characters of the
// walk (read bases pre .. pre + T - 1; the first `at` of them lie in node r) leave the node: one candidate per allele of the
// site behind it -- the allele's bases, then node r + 1 -- and, when that does not hold the rest, per allele of the site
// behind node r + 1 as well; the labels of the candidates with the fewest mismatches (at most `budget`) are the result.
// When those end at ONE position they are one path (labels with equal ends, genotype_paths.cpp:32-66) whose allele sets
// are the union over the candidates (path.cpp:105-129), site r first.
// status: 0 = not this shape (a third site, an allele without a base, a killing character), 1 = `got` mismatches (INVALID:
// no candidate within the budget), `end`, the sets; 2 = the best candidates end at different positions (several paths)
struct HintWalk
{
  uint32_t status, got, end, mask1, mask2;
};

template <class Row>
GTX_DEV HintWalk hint_tail_walk(GraphView const & g, IndexView const & ix, Row row, uint32_t idx, uint32_t pre, uint32_t T, uint32_t at,
                                uint32_t r, uint32_t budget)
{
  HintWalk w{0u, INVALID, 0u, 0u, 0u};
  if (r + 1 >= g.n_ref)
    return w;
  uint32_t const nv1 = g.ref_nvar[r], fv1 = g.ref_first_var[r];
  if (nv1 < 2 || nv1 > HINT_MASK_BITS)
    return w;
  uint32_t const n1 = g.ref_len[r + 1], o1 = g.ref_order[r + 1], rest = T - at;
  bool const more = r + 2 < g.n_ref;
  uint32_t const nv2 = more ? g.ref_nvar[r + 1] : 0u, fv2 = more ? g.ref_first_var[r + 1] : 0u;
  uint32_t const n2 = more ? g.ref_len[r + 2] : 0u, o2 = more ? g.ref_order[r + 2] : 0u;
  uint32_t const common = hint_mm_linear(row, ix.refp, pre, idx + pre, at);
  bool kill = false, tie = false, odd = false;
  auto candidate = [&](uint32_t mm, uint32_t end, uint32_t k1, uint32_t k2)
  {
    if (mm > budget)
      return;
    if (mm < w.got)
    {
      w.got = mm;
      w.end = end;
      w.mask1 = k1;
      w.mask2 = k2;
      tie = false;
      return;
    }
    if (mm == w.got)
    {
      tie = tie || end != w.end || (k2 == 0) != (w.mask2 == 0);
      w.mask1 |= k1;
      w.mask2 |= k2;
    }
  };
#pragma unroll 1
  for (uint32_t a = 0; a < nv1; ++a)
    {
      uint32_t const vl = g.var_len[fv1 + a], vo = g.var_dna[fv1 + a];
      if (vl == 0)
        return w;
      uint32_t mm = common + hint_mm_allele(g, row, pre + at, vo, vl < rest ? vl : rest, kill);
      if (rest <= vl)
      {
        candidate(mm, g_special_of(g, r, g.var_order[fv1 + a] + rest - 1u), 1u << a, 0u);
        continue;
      }
      uint32_t const rem = rest - vl;
      mm += hint_mm_linear(row, ix.refp, pre + at + vl, o1 - g.first_order, rem < n1 ? rem : n1);
      if (rem <= n1)
      {
        candidate(mm, o1 + rem - 1u, 1u << a, 0u);
        continue;
      }
      if (mm > budget) // (the reference drops the candidate here: count_mismatches against the budget, graph.cpp:1268)
        continue;
      if (nv2 < 2 || nv2 > HINT_MASK_BITS)
      {
        odd = true;
        continue;
      }
      uint32_t const rest2 = rem - n1, ro2 = pre + at + vl + n1;
#pragma unroll 1
      for (uint32_t b = 0; b < nv2; ++b)
        {
          uint32_t const vl2 = g.var_len[fv2 + b];
          if (vl2 == 0)
          {
            odd = true;
            continue;
          }
          uint32_t mm2 = mm + hint_mm_allele(g, row, ro2, g.var_dna[fv2 + b], vl2 < rest2 ? vl2 : rest2, kill);
          if (rest2 <= vl2)
          {
            candidate(mm2, g_special_of(g, r + 1, g.var_order[fv2 + b] + rest2 - 1u), 1u << a, 1u << b);
            continue;
          }
          uint32_t const rem2 = rest2 - vl2;
          if (rem2 > n2) // a third site
          {
            odd = odd || mm2 <= budget;
            continue;
          }
          mm2 += hint_mm_linear(row, ix.refp, ro2 + vl2, o2 - g.first_order, rem2);
          candidate(mm2, o2 + rem2 - 1u, 1u << a, 1u << b);
        }
    }
  if (odd || kill)
    return w;
  w.status = tie ? 2u : 1u;
  return w;
}

// The forward task of one read.  Returns true when the record was written, false = declined (nothing written).
// `row`: the read in plane form as words (global memory, or the copy the kernel staged in LDS).
// `stage` (may be NULL): room for HINT_STAGE_WORDS words; a record that fits is written there instead (zeros behind its
// end) and the caller moves it to its slot -- the kernel does that four lanes per record.  Returns 0 = declined, 1 = the
// record is in `rec`, 2 = it is in `stage`, HINT_TO_GENERAL = declined and known to be declined by the express pass too.
constexpr uint32_t HINT_STAGE_WORDS = 16; // a record of up to three variant sites (6 + 3 * 3 words)
constexpr uint32_t HINT_TO_GENERAL = 3;

// DENSE: the build for graphs whose sites lie close together (k-mers over two sites, walks over sites with alleles of any
// length, allele windows): more registers, the same records where both builds finish a read.
// hinted_on_path: the read against the path that table position `idx` lies on -- the linear reference, or (pw != NULL, dense
// build) the allele window *pw, whose position 0 is table position `wbase`.  mm_all: the compare's mismatches over the read.
// where a record that is not staged goes: a pointer, or something that yields it when asked (the kernel: the slot's address is
// worked out where it is needed -- two registers less across the whole read)
template <class R>
GTX_DEV uint32_t * hint_rec_ptr(R const & r)
{
  if constexpr (std::is_pointer<R>::value)
    return r;
  else
    return r();
}

template <bool DENSE, class Row, class Rec>
GTX_DEV uint32_t hinted_on_path(GraphView const & g, IndexView const & ix, Row row, uint32_t seq_stride, gtx_read_meta const & m, uint32_t idx,
                                HintWindow const * pw, uint32_t wbase, Rec rec_at, uint32_t rec_words, uint32_t * stage, uint32_t & mm_all)
{
  uint32_t const L = m.l_qseq;
  // graph order of table position p (of this path)
  auto order_of = [&](uint32_t p) -> uint32_t
  {
    if constexpr (DENSE)
      if (pw)
        return hint_win_order(g, *pw, p - wbase);
    return g.first_order + p;
  };
  uint32_t const n_k = 1 + (L - K) / (K - 1);
  // ---- the read and the reference under it, 8 bases per word, aligned to the read
  HintCounts h{};
  uint32_t const * refw = ix.refp + 4 * (idx >> 5);
  uint32_t const sh = idx & 31u;
  // ---- the flags of the k-mers' places (their second words also describe the positions the walks start from); all
  //      issued together with the reference words: one round trip
  uint2_t const f0 = ix.pos_flags[idx], f1 = ix.pos_flags[idx + (K - 1)];
  uint2_t const f2 = ix.pos_flags[idx + (n_k > 2 ? 2 * (K - 1) : 0u)], f3 = ix.pos_flags[idx + (n_k > 3 ? 3 * (K - 1) : 0u)];
  uint2_t const f4 = ix.pos_flags[idx + (n_k > 4 ? 4 * (K - 1) : 0u)];
  uint32_t const y_end = ix.pos_flags[idx + (K - 1) * n_k].y; // the position behind the last k-mer (31 n_k <= L - 1: inside the read)
  uint2_t const t_end = ix.tail_info[idx + (K - 1) * n_k];    // ... and the site behind its reference node
  GTX_X_STOP(0, f0.x ^ f1.y ^ f2.x ^ f3.y ^ f4.x ^ y_end ^ t_end.x ^ t_end.y ^ refw[0]);
  hint_compare(row, seq_stride, refw, sh, L, h);
  mm_all = hc_all(h);
  GTX_X_STOP(1, h.k[0] ^ h.k[1] ^ h.k[2] ^ h.k[3] ^ h.k[4] ^ h.upto ^ h.more ^ f0.x ^ f1.y ^ f2.x ^ f3.y ^ f4.x ^ y_end ^ t_end.x);
  // ---- every k-mer: the label of its place, no label at all, or not provable
  uint32_t const none = hk_make(HINT_K_HOLE, HINT_NO_SITE, 0u, false, false);
  uint32_t amb2 = 0; // k-mers with one ambiguous base in each half: three more filter probes (below)
  uint32_t k0 = hint_kmer<0, DENSE>(f0, row, h, amb2), k1 = hint_kmer<1, DENSE>(f1, row, h, amb2);
  uint32_t k2 = n_k > 2 ? hint_kmer<2, DENSE>(f2, row, h, amb2) : none;
  uint32_t k3 = n_k > 3 ? hint_kmer<3, DENSE>(f3, row, h, amb2) : none;
  uint32_t k4 = n_k > 4 ? hint_kmer<4, DENSE>(f4, row, h, amb2) : none;
  if ((k0 & 3u) == HINT_K_DECLINE || (k1 & 3u) == HINT_K_DECLINE || (k2 & 3u) == HINT_K_DECLINE || (k3 & 3u) == HINT_K_DECLINE ||
      (k4 & 3u) == HINT_K_DECLINE)
    return false;
  GTX_X_STOP(2, k0 ^ (k1 << 1) ^ (k2 << 2) ^ (k3 << 3) ^ (k4 << 4) ^ amb2 ^ h.upto ^ h.more ^ y_end ^ t_end.x);
  if ((k0 | k1 | k2 | k3 | k4) & (HK_NEED_LEFT | HK_NEED_RIGHT))
  {
    // ---- the filter probes.  The pass is bound by the vector instructions a wavefront issues (four cycles each on gfx950: a
    //      wavefront's 1 300 are its whole life with six of them per SIMD -- tools/ubench/valu_rate.hip), and a wavefront executes
    //      what ANY of its lanes asks for: with HINT_NEAR_FREE three lanes in a hundred still hang on a probe, nearly always on
    //      one, but the ten slots written out side by side (k-mer x half, 28 instructions each) ran in every wavefront.  Now a
    //      lane takes its needed halves one after the other, base offset and filter chosen at run time: one slot's instructions
    //      per round, as many rounds as the wavefront's neediest lane has halves (one, rarely two).
    uint32_t need = ((k0 >> 6) & 3u) | (((k1 >> 6) & 3u) << 2) | (((k2 >> 6) & 3u) << 4) | (((k3 >> 6) & 3u) << 6) | (((k4 >> 6) & 3u) << 8);
    static_assert(HK_NEED_LEFT == 64u && HK_NEED_RIGHT == 128u, "bit 2 I + side of `need`");
    bool maybe = false;
#ifndef GTX_X_NO_FILTER /* (experiment build: the kernel without its filter loads -- every probed half "absent") */
    while (need != 0)
    {
      // (two halves a round, their loads in flight together: one half a round made a lane with two needed halves -- one wavefront
      //  in two had such a lane -- wait for two round trips one after the other, and the wavefront with it.  The second half's
      //  slot is worked out by the lanes that have one: a wavefront without such a lane -- most, with filters of 128 bits per
      //  key -- branches round it)
      uint32_t const j0 = static_cast<uint32_t>(__builtin_ctz(need)); // k-mer j / 2, half j % 2
      need &= need - 1u;
      uint32_t a0, a1, word0, mask0, x1 = 0, mask1 = 1u;
      plane_extract16_at(row, (K - 1) * (j0 >> 1) + 16u * (j0 & 1u), a0, a1);
      hint_filter_slot(a0, a1, ix.filt_log2, word0, mask0);
      uint32_t const x0 = ix.filt[j0 & 1u][word0];
      if (need != 0)
      {
        uint32_t const j1 = static_cast<uint32_t>(__builtin_ctz(need));
        need &= need - 1u;
        uint32_t b0, b1, word1;
        plane_extract16_at(row, (K - 1) * (j1 >> 1) + 16u * (j1 & 1u), b0, b1);
        hint_filter_slot(b0, b1, ix.filt_log2, word1, mask1);
        x1 = ix.filt[j1 & 1u][word1];
      }
      maybe = maybe || (x0 & mask0) == mask0 || (x1 & mask1) == mask1;
    }
#endif
    if (maybe)
    {
      GTX_HINT_NOTE(4); // a half that has to be absent may occur in the index (a variant allele, or a filter collision)
      return false;
    }
  }
  if ((k0 & 3u) == HINT_K_DECLINE || (k1 & 3u) == HINT_K_DECLINE || (k2 & 3u) == HINT_K_DECLINE || (k3 & 3u) == HINT_K_DECLINE ||
      (k4 & 3u) == HINT_K_DECLINE)
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
      uint32_t const lo_w = row[4 * w + b], hi_w = row[4 * (w + 1 < HINT_PLANE_WORDS ? w + 1 : w) + b];
      p[b] = hint_funnel(lo_w, hi_w, sft) & 0xFFFFu; // bases A .. A+15 (A + 15 < 160: inside the row's five groups)
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
  GTX_X_STOP(3, k0 ^ (k1 << 1) ^ (k2 << 2) ^ (k3 << 3) ^ (k4 << 4) ^ amb2 ^ h.upto ^ h.more ^ y_end ^ t_end.x);
  auto bits = [&](uint32_t flag, uint32_t want) // one bit per k-mer
  {
    return ((k0 & flag) == want ? 1u : 0u) | ((k1 & flag) == want ? 2u : 0u) | ((k2 & flag) == want ? 4u : 0u) | ((k3 & flag) == want ? 8u : 0u) |
           ((k4 & flag) == want ? 16u : 0u);
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
    // (the runs of the five-bit word by its shifted conjunctions -- bit i of x_m: k-mers i .. i + m are all labelled -- instead
    //  of a walk over the k-mers with the best and the second-best run carried along: 20 instructions for 90, in every
    //  wavefront that has a lane with a label-less k-mer, which is nearly every one.  Bits at and above n_k are clear: those
    //  k-mers were given the verdict `none`.)
    uint32_t const x0 = labelled, x1 = x0 & (x0 >> 1), x2 = x1 & (x1 >> 1), x3 = x2 & (x2 >> 1), x4 = x3 & (x3 >> 1);
    static_assert(AlignCfg::KC == 5, "runs of up to five k-mers");
    uint32_t const deepest = x4 ? x4 : x3 ? x3 : x2 ? x2 : x1 ? x1 : x0; // one bit per longest run, at its first k-mer
    uint32_t const best_len = (x0 != 0 ? 1u : 0u) + (x1 != 0 ? 1u : 0u) + (x2 != 0 ? 1u : 0u) + (x3 != 0 ? 1u : 0u) + (x4 != 0 ? 1u : 0u);
    uint32_t const n_best = static_cast<uint32_t>(__builtin_popcount(deepest));
    uint32_t best_lo = deepest ? static_cast<uint32_t>(__builtin_ctz(deepest)) : 0u;                  // the first of the longest runs ...
    uint32_t const second_lo = deepest ? 31u - static_cast<uint32_t>(__builtin_clz(deepest)) : 0u; // ... and the last one (looked at when there are two)
    uint32_t const second = (best_len == 0 || n_best >= 2) ? best_len : 0u; // (only "as long as the best one" is asked below)
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
      if (best_len == 0 || n_best != 2 || (f0.y & 255u) < L || (par & (run_a | run_b)) != 0)
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
  GTX_X_STOP(4, lo ^ (hi << 3) ^ (mism << 6) ^ (two_rs << 9) ^ (two_re << 17) ^ (two_mism << 25) ^ k0 ^ k1 ^ k2 ^ k3 ^ k4 ^ h.upto ^ h.more ^ y_end ^ t_end.x ^ static_cast<uint32_t>(decided) ^ static_cast<uint32_t>(par_start));
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
    uint32_t set = (km >> HK_SET_SHIFT) & 255u;
    if (DENSE && hk_is_two(km))
    {
      uint32_t const s0 = km >> HK_SITE_SHIFT;
      if (site != s0 && site != s0 + 1u)
        return 4u;
      set = site == s0 ? (set & 15u) : (set >> 4);
      return (set & (set - 1u)) != 0 ? 4u : static_cast<uint32_t>(__builtin_ctz(set));
    }
    if ((km >> HK_SITE_SHIFT) != site || (set & (set - 1u)) != 0)
      return 4u;
    return set ? static_cast<uint32_t>(__builtin_ctz(set)) : (km >> HK_ALLELE_SHIFT) & 3u;
  };
  uint32_t head_site = 0, head_mask = 0; // the site the walk at the read's start crossed, with its best alleles
  if (prs != 0 && !decided) // walk_read_starts (genotype_paths.cpp:555-621)
  {
    uint32_t const y = lo == 1 ? f1.y : lo == 2 ? f2.y : lo == 3 ? f3.y : f4.y; // (position 31 lo is k-mer lo's own place)
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
      uint32_t const km = lo == 1 ? k1 : lo == 2 ? k2 : lo == 3 ? k3 : k4;
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
  uint32_t tail_mask2 = 0;               // ... and (dense build) those of the site behind it, when the walk crossed that as well
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
    uint32_t tail_end = DENSE ? order_of(idx + L - 1u) : 0u; // (dense build; inside one reference node, or over SNP-like sites: the path's own position)
    uint32_t const y = hi + 1 == n_k ? y_end : hi == 0 ? f1.y : hi == 1 ? f2.y : hi == 2 ? f3.y : f4.y;
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
        only = carried(hi == 0 ? k0 : hi == 1 ? k1 : hi == 2 ? k2 : hi == 3 ? k3 : k4, ti.y);
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
        bool walked = false;
        if constexpr (DENSE)
          if (!on_site && (ti.x & HINT_TAIL_NODE) != 0 && tail_len <= 64)
          {
            HintWalk const w = hint_tail_walk(g, ix, row, idx, pre, tail_len, at, ti.y, budget);
            if (w.status == 2)
            {
              GTX_HINT_NOTE(8);
              return HINT_TO_GENERAL;
            }
            if (w.status == 1)
            {
              walked = true;
              got = w.got; // (INVALID: no candidate within the budget -- the path stays as it is)
              tail_end = w.end;
              tail_site = ti.y;
              tail_mask = w.mask1;
              tail_mask2 = w.mask2;
            }
          }
        if (!walked)
        {
          GTX_HINT_NOTE(8);
          return false;
        }
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
      if constexpr (DENSE)
        end = tail_end;
      else
        end += tail_len - 1;
    }
    else
      tail_mask = tail_mask2 = 0; // (the path stays as it is: no site from the walk)
  }
  GTX_X_STOP(5, start ^ (end << 1) ^ (rs << 3) ^ (re << 11) ^ (mism << 20) ^ tail_site ^ tail_mask ^ tail_mask2 ^ head_site ^ head_mask ^ k0 ^ k1 ^ k2 ^ k3 ^ k4);
  // ---- variant sites of the path, most recent k-mer first (Path(p1, p2), path.cpp:38-82); a site under two
  //      neighbouring k-mers is one entry (the same base, hence the same allele)
  uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0; // site << 16 | allele mask (named registers: an indexed array would live in scratch)
  uint32_t nvar = 0;
  uint32_t last = 0xFFFFFFFFu;
  bool clash = false;
  auto append = [&](uint32_t entry)
  {
    if constexpr (DENSE)
    {
      // (a k-mer may bring two sites: the list is not in the sites' order any more -- Path(p1, p2) looks a site of p1 up among
      //  all of p2's, path.cpp:38-82)
      bool found = false;
      auto meet = [&](uint32_t & v, uint32_t at)
      {
        if (at < nvar && (v >> 16) == (entry >> 16))
        {
          v &= entry | 0xFFFF0000u;
          clash = clash || (v & 0xFFFFu) == 0;
          found = true;
        }
      };
      meet(v0, 0);
      meet(v1, 1);
      meet(v2, 2);
      meet(v3, 3);
      meet(v4, 4);
      meet(v5, 5);
      if (!found)
      {
        v0 = nvar == 0 ? entry : v0;
        v1 = nvar == 1 ? entry : v1;
        v2 = nvar == 2 ? entry : v2;
        v3 = nvar == 3 ? entry : v3;
        v4 = nvar == 4 ? entry : v4;
        v5 = nvar == 5 ? entry : v5;
        ++nvar;
      }
    }
    else if ((entry >> 16) == (last >> 16))
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
      if (DENSE && hk_is_two(km))
      {
        append(((km >> HK_SITE_SHIFT) << 16) | (set & 15u));
        append((((km >> HK_SITE_SHIFT) + 1u) << 16) | (set >> 4));
      }
      else
        append(((km >> HK_SITE_SHIFT) << 16) | (set ? set : 1u << ((km >> HK_ALLELE_SHIFT) & 3u)));
    }
  };
  if (tail_mask != 0) // (the walk's labels are merged last: their site comes first)
    append((tail_site << 16) | tail_mask);
  if (tail_mask2 != 0)
    append(((tail_site + 1u) << 16) | tail_mask2);
  push(4, k4);
  push(3, k3);
  push(2, k2);
  push(1, k1);
  push(0, k0);
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
    uint32_t const amb_lo = hc_get(lo == 1 ? h.k[1] : lo == 2 ? h.k[2] : lo == 3 ? h.k[3] : h.k[4], HC_AMB);
    uint2_t const f_lo = lo == 1 ? f1 : lo == 2 ? f2 : lo == 3 ? f3 : f4;
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
      uint32_t const mm_run = mmk & run, km_hi = hi == 0 ? k0 : hi == 1 ? k1 : hi == 2 ? k2 : hi == 3 ? k3 : k4;
      auto plain = [&](uint32_t k, uint32_t km) { return ((run >> k) & 1u) == 0 || (km & ((3u << HK_ALLELE_SHIFT) | (255u << HK_SET_SHIFT))) == 0; };
      if (!(plain(0, k0) && plain(1, k1) && plain(2, k2) && plain(3, k3) && plain(4, k4)))
      {
        GTX_HINT_NOTE(10);
        return HINT_TO_GENERAL;
      }
      bool const region_clean = hc_upto(h, hi + 1) == hc_upto(h, lo);
      if (mm_run == (1u << hi) && hi + 1 > 4)
      {
        // (the fifth k-mer of a read of 156 bases and more: whether its substitution sits on its last base -- base 155, the
        //  tail walk's first -- is not among the compare's counts)
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
  uint32_t * rec = to_stage ? stage : hint_rec_ptr(rec_at);
  if (to_stage)
  {
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

// The forward task of one read (see hinted_on_path for the return values): on the linear reference at the hinted place, and --
// dense build, when that is declined -- on the window of the allele the read seems to carry: of the alternative alleles of
// the (up to four) sites under the read that have windows, the one whose path the read differs least from, if that is less
// than it differs from the linear reference.  Whatever path is tried, a record is only written when every lookup on it is proven.
template <bool DENSE, class Row, class Rec>
GTX_DEV uint32_t hinted_one(GraphView const & g, IndexView const & ix, Row row, uint32_t seq_stride, gtx_read_meta const & m,
                            Rec rec, uint32_t rec_words, uint32_t * stage = nullptr)
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
  uint32_t mm_main = 0xFFFFu;
  uint32_t where = hinted_on_path<DENSE>(g, ix, row, seq_stride, m, idx, static_cast<HintWindow const *>(nullptr), 0u, rec, rec_words, stage, mm_main);
  if constexpr (DENSE)
  {
    if (where == 0 && ix.n_win != 0 && mm_main != 0xFFFFu)
    {
      uint2_t const t0 = ix.tail_info[idx];
      if ((t0.x & HINT_TAIL_NODE) != 0)
      {
        uint32_t best = mm_main, best_at = 0, best_w = INVALID, tried = 0;
        uint32_t r = t0.y;
#pragma unroll 1
        for (uint32_t s = 0; s < 4 && r + 1 < g.n_ref; ++s, ++r)
        {
          uint32_t const off = g.ref_order[r] + g.ref_len[r] - (g.first_order + idx); // the read's base on the site's first position
          if (off >= L)
            break;
          uint32_t const sw = ix.site_win[r], first = sw & 0xFFFFFFu, count = sw >> 24;
#pragma unroll 1
          for (uint32_t k = 0; k < count && tried < 8; ++k, ++tried)
          {
            uint32_t const at = ix.win_base + (first + k) * HINT_WIN_STRIDE + (HINT_WIN_BEFORE - off);
            uint32_t const mm = hint_mm_linear(row, ix.refp, 0u, at, L);
            if (mm < best)
            {
              best = mm;
              best_at = at;
              best_w = first + k;
            }
          }
        }
        if (best_w != INVALID)
        {
          uint32_t mm_w = 0;
          where = hinted_on_path<DENSE>(g, ix, row, seq_stride, m, best_at, ix.win + best_w, ix.win_base + best_w * HINT_WIN_STRIDE, rec, rec_words, stage, mm_w);
        }
      }
    }
  }
  return where;
}

} // namespace gtx
