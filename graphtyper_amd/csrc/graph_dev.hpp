// graph_dev.hpp -- device-side helpers that do not depend on the kernel's table sizes: graph / special-position
// lookups, the bucketed index tables, half keys, wave-policy macros.  (Split from align_core: see align_core.hpp.)
//
// What the reference does per read (src/typer/alignment.cpp:23-103, find_genotype_paths_of_one_of_the_sequences) with
// heap containers, restated over fixed tables in LDS.
//
// Execution style ("wave-uniform + lane lambdas"): all 64 lanes run the same control flow on the same values (state
// lives in LDS, scalars are replicated), LDS writes are done by the leader lane only, and the data-parallel pieces --
// read unpacking, 2-bit key assembly, the 97 index probes per k-mer, stable hit compaction, character comparison of a
// read against graph sequence, table copies -- are expressed as lambdas over the lane index plus wave primitives
// (ballot, exclusive scan).  A policy type W supplies those primitives: WaveHip (gtx_api.hip) maps them to the
// hardware; tests/emu supplies a sequential stand-in so the very same source can be debugged without a GPU.
#pragma once
#include <cstdint>

#include "gtx_flat.hpp"

#if defined(__HIPCC__)
#define GTX_DEV __device__ inline
#else
#define GTX_DEV inline
#endif

namespace gtx
{
// include/graphtyper/constants.hpp.in:43-48
constexpr uint32_t MAX_UNIQUE_KMER_POSITIONS = 512;
constexpr uint32_t MAX_SEED_NUMBER_ALLOWING_MISMATCHES = 64;
constexpr uint32_t MAX_SEED_NUMBER_FOR_WALKING = 256;
constexpr uint32_t MAX_NUM_LOCATIONS_PER_PATH = 256;

// graph sequence is stored as codes: IUPAC letters keep their 4-bit BAM code (A=1 C=2 G=4 T=8 N=15), '<' and '>'
// (SV breakpoint tags, graph_utils.hpp:20-23) become DNA_KILL, anything else DNA_OTHER (equal to no read character)
constexpr uint8_t DNA_KILL = 0x80, DNA_OTHER = 0x40;

struct alignas(16) uint4_t // 16-byte move
{
  uint32_t x, y, z, w;
};

// ---- reads as bit planes (the layout every alignment kernel reads; gtx.h: gtx_align_batch_planes).  A read row of S bytes
// (S a multiple of 16) is S / 16 groups of 32 bases, four words per group: word 4g + b holds bit b of the BAM codes (A=1 C=2
// G=4 T=8, N=15, '='=0) of bases 32g .. 32g+31, base 32g + j at bit j.  The per-base questions of the position-hinted pass
// (differs? one base or a set?) are plain bitwise operations on such words, a 2-bit key half is two shifts and an OR, and
// no kernel transposes anything per read and step: the host (gtx_stream_push) or one repack at staging time does it once.
constexpr uint32_t PLANE_GROUP_BYTES = 16;

#if defined(__HIPCC__)
#define GTX_HDI __host__ __device__ inline
#else
#define GTX_HDI inline
#endif

// BAM code of base i of a plane row
GTX_HDI uint32_t plane_code_at(uint32_t const * row, uint32_t i)
{
  uint32_t const * g = row + 4u * (i >> 5);
  uint32_t const s = i & 31u;
  return ((g[0] >> s) & 1u) | (((g[1] >> s) & 1u) << 1) | (((g[2] >> s) & 1u) << 2) | (((g[3] >> s) & 1u) << 3);
}

// four bits -> four bytes (bit k to bit 0 of byte k): one multiply whose partial products land on distinct bits
GTX_HDI uint32_t plane_spread4(uint32_t nibble)
{
  return (nibble * 0x00204081u) & 0x01010101u;
}

// the codes of bases o .. o+3 (o a multiple of 4) of a group, one per byte
GTX_HDI uint32_t plane_codes4(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t o)
{
  return plane_spread4((p0 >> o) & 15u) | (plane_spread4((p1 >> o) & 15u) << 1) | (plane_spread4((p2 >> o) & 15u) << 2) |
         (plane_spread4((p3 >> o) & 15u) << 3);
}

// bit B of the 8 bases of a little-endian word of 4 BAM bytes (byte i holds base 2i in its high and base 2i+1 in its low
// nibble) as a byte, base j at bit j: the four bits `B` of the nibbles of a 16-bit half are gathered by ONE multiply whose
// partial products do not collide.  hi = w >> 16.
template <uint32_t B>
GTX_HDI uint32_t nib_plane_byte(uint32_t w, uint32_t hi)
{
  // source bits of a half, by base: base 1 at bit B, base 0 at 4+B, base 3 at 8+B, base 2 at 12+B; they go to 12+B .. 15+B
  constexpr uint32_t M = (1u << 13) | (1u << 8) | (1u << 7) | (1u << 2);
  uint32_t const pl = (w & (0x1111u << B)) * M, ph = (hi & (0x1111u << B)) * M;
  return ((pl >> (12 + B)) & 0xFu) | ((ph >> (8 + B)) & 0xF0u);
}

// one group of a plane row from (up to) 16 bytes of a BAM nibble row: w[k] = bytes 4k .. 4k+3 as a little-endian word (0
// behind the end of the nibble row)
GTX_HDI void planes_from_nibble_words(uint32_t const (&w)[4], uint32_t (&out)[4])
{
  out[0] = out[1] = out[2] = out[3] = 0;
  for (uint32_t k = 0; k < 4; ++k)
  {
    uint32_t const hi = w[k] >> 16;
    out[0] |= nib_plane_byte<0>(w[k], hi) << (8 * k);
    out[1] |= nib_plane_byte<1>(w[k], hi) << (8 * k);
    out[2] |= nib_plane_byte<2>(w[k], hi) << (8 * k);
    out[3] |= nib_plane_byte<3>(w[k], hi) << (8 * k);
  }
}

// a whole row: nibble row of `nib_bytes` bytes -> `groups` groups (bases beyond the nibble row: code 0)
GTX_HDI void planes_from_nibbles(uint8_t const * nib, uint32_t nib_bytes, uint32_t * out, uint32_t groups)
{
  for (uint32_t g = 0; g < groups; ++g)
  {
    uint32_t w[4];
    for (uint32_t k = 0; k < 4; ++k)
    {
      uint32_t v = 0;
      for (uint32_t b = 0; b < 4; ++b)
      {
        uint32_t const at = 16u * g + 4u * k + b;
        v |= (at < nib_bytes ? static_cast<uint32_t>(nib[at]) : 0u) << (8 * b);
      }
      w[k] = v;
    }
    uint32_t o[4];
    planes_from_nibble_words(w, o);
    out[4 * g + 0] = o[0];
    out[4 * g + 1] = o[1];
    out[4 * g + 2] = o[2];
    out[4 * g + 3] = o[3];
  }
}

// internal status bit (never stored in a record): the task met an allele number beyond the allele sets of the pass it was
// in; together with GTX_ST_PATH_OVERFLOW it sends the task on, in the end to the pass with GTX_WIDE_MASK_WORDS-word sets
constexpr uint32_t GTX_ST_WIDE_ALLELE = 32u;

// Wave-uniform sections that write shared state.  Every lane holds the same values there, so on the device all lanes
// simply execute the section -- 64 identical stores to one address are one store -- which saves the exec-mask save /
// restore around each of them (the general pass is bound by scalar instructions: one per cycle and CU).  The emulation
// (and the profiling build, whose counters are incremented inside such sections) runs them on the leader only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GTX_PROF)
#define GTX_LEAD
#else
#define GTX_LEAD if (W::leader())
#endif

// Values that are equal on all lanes by construction (loaded from LDS state or from graph tables at a uniform address)
// are moved to scalar registers: control flow on them then compiles to scalar branches instead of exec-mask juggling.
#define GTX_U(x) W::uni(x)

// phase timing (profiling build only: make -C graphtyper_amd/csrc prof -> libgtx_prof.so)
#ifdef GTX_PROF
// (per-wave sums in the workspace, flushed once when the wave retires: one global atomic per tick would serialise)
#define GTX_PROF_BEGIN unsigned long long _pt = W::clock();
#define GTX_PROF_RESET _pt = W::clock();
#define GTX_PROF_TICK(k)                                   \
  {                                                        \
    unsigned long long const _pn = W::clock();             \
    GTX_LEAD ws.prof_acc[k] += _pn - _pt;                  \
    _pt = W::clock();                                      \
  }
#else
#define GTX_PROF_BEGIN
#define GTX_PROF_RESET
#define GTX_PROF_TICK(k)
#endif
// (a one-off profiling build, -DGTX_PROF -DGTX_PROF_WALK: the walks' parts in slots 2..5 -- 2 the path's geometry and the shortcuts, 3
//  get_locations, 4 iterative_dfs, 5 the lists' copies and their chaining --, what those slots hold otherwise goes to slot 1)
#if defined(GTX_PROF) && defined(GTX_PROF_WALK)
#define GTX_PROF_TICKA(k) GTX_PROF_TICK(1)
#define GTX_PROF_WBEGIN GTX_PROF_BEGIN
#define GTX_PROF_WTICK(k) GTX_PROF_TICK(k)
#else
#define GTX_PROF_TICKA(k) GTX_PROF_TICK(k)
#define GTX_PROF_WBEGIN
#define GTX_PROF_WTICK(k)
#endif

// ---------------------------------------------------------------------------------------------------------------
// graph helpers
// ---------------------------------------------------------------------------------------------------------------
GTX_DEV bool g_is_special(GraphView const & g, uint32_t pos)
{
  return pos >= SPECIAL_START && (pos - SPECIAL_START) < g.n_special; // graph.cpp:1784-1787
}

GTX_DEV uint32_t g_ref_reach_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? g.special_ref_reach[pos - SPECIAL_START] : pos; // graph.cpp:1789-1795
}

GTX_DEV uint32_t g_actual_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? g.special_actual[pos - SPECIAL_START] : pos; // graph.cpp:1797-1803
}

// Graph::get_special_pos for a position inside an allele of `site` (graph.cpp:1775-1782); identity up to the
// reference allele's reach.
GTX_DEV uint32_t g_special_of(GraphView const & g, uint32_t site, uint32_t pos)
{
  uint32_t const rr = g.site_ref_reach[site];
  return pos > rr ? SPECIAL_START + g.site_special_base[site] + (pos - rr - 1) : pos;
}

// graph order of position `local` of a window: the linear reference in front of the site, the allele (special positions
// beyond the reference allele's reach, graph.cpp:1775-1782), the linear reference behind the site's reference allele
GTX_HDI uint32_t hint_win_order(GraphView const & g, HintWindow const & w, uint32_t local)
{
  if (local < HINT_WIN_BEFORE)
    return w.site_order - (HINT_WIN_BEFORE - local);
  uint32_t const k = local - HINT_WIN_BEFORE;
  if (k < w.len_a)
  {
    uint32_t const pos = w.site_order + k, rr = g.site_ref_reach[w.site];
    return pos > rr ? SPECIAL_START + g.site_special_base[w.site] + (pos - rr - 1) : pos;
  }
  return w.site_order + w.len_0 + (k - w.len_a);
}

GTX_DEV uint32_t site_order(GraphView const & g, uint32_t site)
{
  return g.ref_order[site] + g.ref_len[site]; // order of the site's variant nodes
}

// wave-uniform variants of the helpers above (arguments uniform, results in scalar registers)
template <class W>
GTX_DEV uint32_t ug_ref_reach_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? GTX_U(g.special_ref_reach[pos - SPECIAL_START]) : pos;
}

template <class W>
GTX_DEV uint32_t ug_actual_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? GTX_U(g.special_actual[pos - SPECIAL_START]) : pos;
}

template <class W>
GTX_DEV uint32_t ug_special_of(GraphView const & g, uint32_t site, uint32_t pos)
{
  uint32_t const rr = GTX_U(g.site_ref_reach[site]);
  return pos > rr ? SPECIAL_START + GTX_U(g.site_special_base[site]) + (pos - rr - 1) : pos;
}

template <class W>
GTX_DEV uint32_t ug_site_order(GraphView const & g, uint32_t site)
{
  return GTX_U(g.ref_order[site]) + GTX_U(g.ref_len[site]);
}

// last reference node whose order is <= pos (the `rr` of graph.cpp:950-955); pos >= first_order required
template <class W>
GTX_DEV uint32_t g_ref_node_at(GraphView const & g, uint32_t pos)
{
  if (g.pos_node && pos - g.first_order < g.n_pos_info)
  {
    // the position table answers in one load when the position lies inside a reference node (the usual case)
    uint32_t const direct = GTX_U(g.pos_node[pos - g.first_order]);
    if (direct != INVALID)
      return direct;
  }
  uint32_t b = (pos - g.first_order) >> POS_BUCKET_SHIFT;
  if (b >= g.n_bucket)
    b = g.n_bucket - 1;
  uint32_t r = GTX_U(g.pos_bucket[b]);
  while (r + 1 < g.n_ref && GTX_U(g.ref_order[r + 1]) <= pos)
    ++r;
  return r;
}

// word-wise LDS -> LDS copy of a table entry, one word per lane
template <class W, class T>
GTX_DEV void copy_entry(T & dst, T const & src)
{
  static_assert(sizeof(T) % 4 == 0, "entries are copied word-wise");
  if (&dst == &src)
    return;
  uint32_t * d = reinterpret_cast<uint32_t *>(&dst);
  uint32_t const * s = reinterpret_cast<uint32_t const *>(&src);
  W::lanes([&](uint32_t l) {
    if constexpr (sizeof(T) / 4 <= 64)
    {
      if (l < sizeof(T) / 4)
        d[l] = s[l];
    }
    else
      for (uint32_t w = l; w < sizeof(T) / 4; w += 64) // the second pass' entries are longer than one wave
        d[w] = s[w];
  });
  W::lds_sync();
}

// small fixed bit set held in registers (one word for the main pass' table sizes)
template <uint32_t N>
struct BitSet
{
  static constexpr uint32_t WORDS = (N + 63) / 64;
  uint64_t w[WORDS];
  GTX_DEV BitSet()
  {
    for (uint32_t i = 0; i < WORDS; ++i)
      w[i] = 0;
  }
  GTX_DEV void set(uint32_t i)
  {
    if constexpr (WORDS == 1)
      w[0] |= 1ull << i;
    else
      for (uint32_t k = 0; k < WORDS; ++k) // no dynamic indexing: keeps the words in registers
        w[k] |= (k == (i >> 6)) ? (1ull << (i & 63u)) : 0ull;
  }
  GTX_DEV bool get(uint32_t i) const
  {
    if constexpr (WORDS == 1)
      return (w[0] >> i) & 1ull;
    else
    {
      uint64_t v = 0;
      for (uint32_t k = 0; k < WORDS; ++k)
        v |= (k == (i >> 6)) ? w[k] : 0ull;
      return (v >> (i & 63u)) & 1ull;
    }
  }
  GTX_DEV bool any() const
  {
    uint64_t v = 0;
    for (uint32_t i = 0; i < WORDS; ++i)
      v |= w[i];
    return v != 0;
  }
  GTX_DEV uint32_t count() const
  {
    uint32_t c = 0;
    for (uint32_t i = 0; i < WORDS; ++i)
      c += static_cast<uint32_t>(__builtin_popcountll(w[i]));
    return c;
  }
};

// (the same two operations exist for sets that live in memory -- align_core.inl, MemBits --, hence the free functions)
template <class W, uint32_t N>
GTX_DEV void bits_set(BitSet<N> & b, uint32_t i)
{
  b.set(i);
}
template <class W, uint32_t N>
GTX_DEV bool bits_get(BitSet<N> const & b, uint32_t i)
{
  return b.get(i);
}
template <class W, uint32_t N>
GTX_DEV uint64_t bits_word(BitSet<N> const & b, uint32_t k)
{
  uint64_t v = 0;
  for (uint32_t w = 0; w < BitSet<N>::WORDS; ++w)
    v |= w == k ? b.w[w] : 0ull;
  return v;
}
template <class W, uint32_t N>
GTX_DEV void bits_set_word(BitSet<N> & b, uint32_t k, uint64_t mask)
{
  for (uint32_t w = 0; w < BitSet<N>::WORDS; ++w)
    b.w[w] = w == k ? mask : b.w[w];
}

// lookup in a bucketed table (gtx_flat.hpp: BUCKET_SLOTS): the whole 128-byte bucket is one cache line; `hit` (may be
// NULL) receives the slot that matched, whose inline payload is then an L1 hit
GTX_DEV void bucket_find(IndexSlot const * slots, uint32_t log2_buckets, uint64_t key, uint32_t & off, uint32_t & cnt,
                         IndexSlot const ** hit = nullptr, uint32_t * flags = nullptr)
{
  uint64_t const mask = (1ull << log2_buckets) - 1;
  for (uint64_t b = hash_key(key, log2_buckets);; b = (b + 1) & mask)
  {
    IndexSlot const * p = slots + b * BUCKET_SLOTS;
    uint64_t const k0 = p[0].key, k1 = p[1].key, k2 = p[2].key, k3 = p[3].key;
    uint32_t const c0 = p[0].cnt, c1 = p[1].cnt, c2 = p[2].cnt, c3 = p[3].cnt;
    bool const m0 = c0 != 0 && k0 == key, m1 = c1 != 0 && k1 == key;
    bool const m2 = c2 != 0 && k2 == key, m3 = c3 != 0 && k3 == key;
    uint32_t const m = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;
    bool const any = m0 || m1 || m2 || m3;
    uint32_t const raw_off = any ? p[m].off : 0u; // (with the flag bit of gtx_flat.hpp: SLOT_NB_KNOWN)
    off = raw_off & SLOT_OFF_MASK;
    cnt = m0 ? c0 : m1 ? c1 : m2 ? c2 : m3 ? c3 : 0u;
    if (hit)
      *hit = any ? p + m : nullptr;
    if (flags)
      *flags = raw_off & ~SLOT_OFF_MASK;
    // slots fill front to back: an empty last slot means nothing ever spilled out of this bucket
    if (any || c3 == 0)
      return;
  }
}

// PHIndex lookup of one key: (offset, count) of its labels, count 0 when absent
GTX_DEV void index_find(IndexView const & ix, uint64_t key, uint32_t & off, uint32_t & cnt)
{
  bucket_find(ix.slots, ix.log2_cap, key, off, cnt);
}

// bucket of a half key (see IndexView::hslots): (offset, count) into hlist, count 0 when the half does not occur
GTX_DEV void half_find(IndexView const & ix, uint64_t hk, uint32_t & off, uint32_t & cnt)
{
  bucket_find(ix.hslots, ix.h_log2_cap, hk, off, cnt);
}

// table key of the bucket holding every indexed k-mer with the same 16 first (side 0) / last (side 1) bases as q
GTX_DEV uint64_t half_key(uint64_t q, uint32_t side)
{
  uint32_t const lo = static_cast<uint32_t>(q), hi = static_cast<uint32_t>(q >> 32);
  uint64_t const half = side == 0 ? ((lo & 0xFFFFu) | ((hi & 0xFFFFu) << 16)) : ((lo >> 16) | (hi & 0xFFFF0000u));
  return half | (static_cast<uint64_t>(side) << 32);
}

constexpr uint32_t HALF_BUCKET_CAP = 64; // one lane per bucket entry; larger buckets (low-complexity sequence) use the 96 direct probes

// Candidate test shared by both routes below: is `key` at Hamming distance exactly 1 from q, and which neighbour is it?
// (plane-form keys; the reference numbers neighbours by bb = position counted from the LAST base and m = xor of the
// 2-bit code, j = 3*bb + m-1, type_conversions.cpp:272-288)
GTX_DEV bool hamming1_neighbour(uint64_t key, uint64_t q, uint32_t & j)
{
  uint64_t const x = key ^ q;
  uint32_t const d0 = static_cast<uint32_t>(x), d1 = static_cast<uint32_t>(x >> 32);
  uint32_t const bases = d0 | d1; // one bit per differing base
  if (bases == 0 || (bases & (bases - 1)) != 0)
    return false;
  uint32_t const b = static_cast<uint32_t>(__builtin_ctz(bases));
  uint32_t const m = ((d0 >> b) & 1u) | (((d1 >> b) & 1u) << 1);
  j = 3u * (31u - b) + m - 1u;
  return true;
}

// align_read (alignment.cpp:331-363): which orientations a record gets
GTX_DEV bool needs_reverse(gtx_read_meta const & m, bool force_both)
{
  bool const one = (m.flag & 1u) == 0u || (m.tid == m.mtid && m.isize > -1200 && m.isize < 1200 &&
                                           (((m.flag & 16u) != 0u) != ((m.flag & 32u) != 0u)));
  return !one || force_both;
}

} // namespace gtx
