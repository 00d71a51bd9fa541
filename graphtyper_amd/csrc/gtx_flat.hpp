// gtx_flat.hpp -- flat (SoA) graph + k-mer index as the kernels see them, and the host containers that own them.
//
// The reference keeps the graph as vectors of RefNode/VarNode objects with std::vector<char> labels
// (include/graphtyper/graph/graph.hpp:40-134) and the index as phmap::flat_hash_map<uint64_t, std::vector<KmerLabel>>
// (include/graphtyper/index/ph_index.hpp:14-36).  Here both are flat arrays so that one upload puts them in HBM and a
// wavefront can address them with plain offsets.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gtx.h"

namespace gtx
{
constexpr uint32_t K = GTX_K;
constexpr uint32_t INVALID = GTX_INVALID_ID;
constexpr uint32_t SPECIAL_START = GTX_SPECIAL_START;
constexpr uint32_t POS_BUCKET_SHIFT = 6; // position -> ref node table has one entry per 64 bp
constexpr uint32_t MAX_ALLELES = 32u * GTX_WIDE_MASK_WORDS; // MAX_NUMBER_OF_HAPLOTYPES (constants.hpp.in:23); sets of 64 in the front passes

struct alignas(8) uint2_t // 8-byte move
{
  uint32_t x, y;
};

// A k-mer occurrence as stored on the device: KmerLabel (kmer_label.hpp:13-41) with variant_id already resolved to
// (site, allele) -- what Path's constructor derives through Graph::get_variant_order/get_variant_num (path.cpp:13-36).
struct alignas(16) DevLabel
{
  uint32_t start, end;
  uint32_t site;   // ref node index the variant hangs from, INVALID when the label carries no variant
  uint32_t allele; // Graph::get_variant_num()
};

// Hash tables are bucketed: BUCKET_SLOTS consecutive slots (64 bytes) share one hash value, so a lookup is normally one
// 64-byte fetch; a full bucket spills into the next one.  cnt == 0 marks an empty slot.
constexpr uint32_t BUCKET_SLOTS = 4;

// exact table only: bit 31 of a slot's offset word says that every label of every indexed Hamming-1 neighbour of the key lies
// on the interval the key's own labels share and on one of their sites (hint_judge_key) -- what express4's seeding rule asks
// of the neighbours of an exact hit, so that it need not fetch their labels.  Lookups return the offset without it.
// (In the offset, not the count: a lookup tests four counts per bucket and uses one offset.)
constexpr uint32_t SLOT_NB_KNOWN = 0x80000000u, SLOT_OFF_MASK = 0x7FFFFFFFu;

struct alignas(32) IndexSlot
{
  uint64_t key;
  uint32_t off, cnt;
  // What a count of 1 points at, inline: the one label of the key (exact table: start, end, site, allele) or the one
  // entry of the half-key bucket (key low, key high, label offset, label count).  A lookup that ends in a single entry --
  // nearly all of them -- then costs one memory round trip instead of two (the kernels are bound by that latency).
  uint32_t p[4];
};

// Plain-pointer view handed to kernels (device pointers) and to the host emulation in tests (host pointers).
struct GraphView
{
  uint32_t n_ref, n_var, n_special, n_bucket;
  uint32_t first_order; // ref_order[0]
  uint32_t padding;     // back-scan distance of get_locations_of_an_actual_position (graph.cpp:973)
  uint32_t is_sv_graph;
  uint32_t pad0;
  const uint32_t * ref_order;
  const uint32_t * ref_len;
  const uint32_t * ref_dna;
  const uint32_t * ref_nvar;
  const uint32_t * ref_first_var;
  const uint32_t * var_order;
  const uint32_t * var_len;
  const uint32_t * var_dna;
  const uint32_t * var_out_ref;
  const uint32_t * site_ref_reach;    // [n_ref] reach of allele 0 of the site after ref node r (0 when no site)
  const uint32_t * site_special_base; // [n_ref] first special index of the site, INVALID when it has none
  const uint32_t * special_ref_reach; // Graph::ref_reach_poses
  const uint32_t * special_actual;    // Graph::actual_poses
  const uint32_t * pos_bucket;        // [n_bucket] last ref node whose order <= first_order + 64*b
  // [n_pos_info] per position first_order + i: INVALID when it is not inside a reference node, else
  // (offset of its base in `dna` << 8) | min(255, bases of that node from it on); NULL when the arena is > 16 MB
  const uint32_t * pos_info;
  const uint8_t * pos_back; // [n_pos_info] min(255, bases of the position's reference node before it) (backward compares)
  const uint32_t * pos_node; // [n_pos_info] index of the reference node the position lies in (INVALID when in none)
  uint32_t n_pos_info, pad2;
  const char * dna; // graph sequence as codes (see align_core.hpp: DNA_KILL / DNA_OTHER), same offsets as the characters
  // score accumulator layout (haplotype h <-> site h, graph.cpp:680-704)
  const uint64_t * tri_off;    // [n_ref] offset of the genotype triangle of site r
  const uint64_t * allele_off; // [n_ref]
  uint64_t total_tri, total_allele;
  uint32_t n_hap, pad1;
  // connections between near sites (hts_parallel_reader.cpp:800-801 reads pairs less than 100 positions apart): site r's
  // window is the sites r+1 .. near_last[r]; its counters are a [cnum(r)] x [alleles of the window] block at near_off[r]
  const uint32_t * near_last; // [n_ref] (= r when the window is empty)
  const uint64_t * near_off;  // [n_ref]
  uint64_t total_near;
  unsigned long long * prof; // [32] phase cycle counters, only written by the GTX_PROF build (libgtx_prof.so)
};

// One indexed key inside a half-key bucket (see HostIndex::hlist)
struct alignas(16) HalfEntry
{
  uint64_t key;
  uint32_t off, cnt; // its labels
};

constexpr uint32_t HINT_WIN_BEFORE = 160, HINT_WIN_ALLELE_MAX = 64, HINT_WIN_STRIDE = 384; // (160 + 64 + 160: twelve plane groups)
struct HintWindow
{
  uint32_t site, allele, len_a /* bases of the allele */, len_0 /* ... of the site's reference allele */;
  uint32_t site_order /* order of the site's variant nodes */, pad[3];
};

struct IndexView
{
  const IndexSlot * slots;
  const DevLabel * labels;
  uint32_t log2_cap; // log2 of the number of buckets
  uint32_t max_index_labels;
  // half-key (pigeonhole) tables for the Hamming-1 lists: a key within Hamming distance 1 of q shares q's left or
  // right 16 bases exactly, so the 96 neighbour probes of the reference become two bucket lookups
  const IndexSlot * hslots; // key = half | side << 32 ; off/cnt into hlist
  const HalfEntry * hlist;
  uint32_t h_log2_cap;
  uint32_t half_bucket_cap; // buckets above this size use the 96 direct probes instead (0 = always probe directly)
  // ---- position-hinted pass (hinted.hpp).  A read of a sorted BAM comes with the place the mapper put it; the tables
  // below are ordered by reference position, so neighbouring reads share their cache lines, and they carry the PROOF
  // that the global lookups of the reference would return exactly the one label of that place.
  // refp: the linear reference of the region (= the graph's path over every site's allele 0) as four bit planes of its BAM
  //   nibble codes: 4 words (planes 0..3) per 32 positions, position hint_first + 32q + j at bit j of words 4q .. 4q+3;
  //   6 padding groups of zeros.
  // pos_flags[i] (two words), about the 32-mer that starts at hint_first + i (K_i) and about the position itself:
  //  x  HINT_SINGLE_OK  K_i is indexed with exactly the label (i, i+31[, site, allele 0]) (and it may be used: not on a
  //                     variant of an SV graph);
  //     HINT_EXACT_OK   ... and every indexed Hamming-1 neighbour of K_i is that same interval on the same site
  //                     (express4's seeding rule for an exact hit); HINT_PAR: there are such neighbours (the k-mer
  //                     starts a parallel +1-mismatch chain);
  //     HINT_L1 / R1    K_i is the only indexed key with its 16 first / last bases;
  //     HINT_ALT_OK     K_i lies over exactly one site, a SNP (every allele one base, at most 4), and the key of every
  //                     other allele -- K_i with that base replaced -- passes the HINT_EXACT_OK test with its own label
  //                     (i, i+31, site, allele); bits 8..15: the allele number of base A, C, G, T there (2 bits each,
  //                     0 = not an alternative allele); y bits 16..20: the offset of that base in the k-mer;
  //     HINT_MULTI      (with HINT_EXACT_OK, without HINT_SINGLE_OK) K_i lies over a merged site and 2..HINT_OWN_MAX of its
  //                     alleles spell it: that many labels, all (i, i+31) on that site; bits 8..15: the set of their
  //                     allele numbers (all below HINT_MASK_BITS); the neighbours as for HINT_EXACT_OK;
  //     bits 16.. of x  see HINT_SITE_SHIFT: the site K_i's label lies on (HINT_NO_SITE: none);
  //  y  bits 0..7       min(255, bases from this position to the end of its reference node), 0 = not in a reference node;
  //     bits 8..15      min(255, bases of that node in front of the position);
  //     bits 21..26     HINT_SNP_GROUP, reference base and allele count of the SNP under K_i (see below).
  // filt[side]: blocked Bloom filter over every indexed key's 16 first (side 0) / last (side 1) bases in nibble form (four
  //   bits of one word per half): a clear bit proves that no indexed key has that half.
  //   (Round 5, measured and dropped: a first level of an eighth / a quarter / a sixteenth of the words in front of it -- small enough
  //    to stay in an XCD's L2, where the 8 MB of cfg2's two filters are a 64-byte fetch over the fabric per probe, 47 of the 157
  //    bytes the pass fetches per read -- answered 85-97 % of the probes without that fetch and made the pass SLOWER, 0.406 ->
  //    0.425-0.443 ms alone, the step in flight unchanged: its hashes and its second round trip cost more than the bytes saved.
  //    Without any filter load the pass takes 0.372 ms -- the ceiling of anything done to them.)
  // tail_info[i]: the site behind the reference node position i lies in, when a walk from i may cross it the simple way
  //   (a SNP: 2..4 alleles of one base A/C/G/T each; not in an SV graph):  x = HINT_TAIL_OK | alleles << 2 (count, 3 bits) |
  //   min(255, length of the reference node behind the site) << 8 | the alleles' nibble codes << 16 (4 bits each);
  //   y = the site's index = the reference node i lies in (HINT_TAIL_NODE: whatever the site is like; not in an SV graph).
  const uint32_t * refp;
  const uint2_t * pos_flags;
  const uint2_t * tail_info;
  const uint32_t * filt[2];
  uint32_t hint_first, n_hint, filt_log2 /* log2 of the number of words */, pad_hint;
  // ---- allele windows (dense build of the pass).  A read that carries another allele than the reference's at a site does not
  // lie on the linear reference; for the alternative alleles of the sites where that matters (HintWindow) the three tables
  // above continue, behind position win_base, with one window of HINT_WIN_STRIDE positions per (site, allele): the
  // HINT_WIN_BEFORE positions of the linear reference in front of the site, the allele's bases, the linear reference behind the
  // site's reference allele.  A window position's entries mean what a position's of the linear reference mean, with the
  // window's path in place of the linear reference (labels at hint_win_order()).  site_win[r]: first window of the site behind
  // reference node r | number of its windows << 24 (alleles ascending).
  const HintWindow * win;
  const uint32_t * site_win;
  uint32_t win_base, n_win, pad_win[2];
};

// positions the three per-position tables hold: the linear reference, padding up to win_base, the windows, padding
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t hint_win_base(uint32_t n_hint)
{
  return (n_hint / 64u + 5u) * 64u; // (whole plane groups, at least 8 of them behind the linear reference as before)
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t hint_total_positions(uint32_t n_hint, uint32_t n_win)
{
  return n_win == 0 ? n_hint : static_cast<uint64_t>(hint_win_base(n_hint)) + static_cast<uint64_t>(n_win) * HINT_WIN_STRIDE + 256u;
}

constexpr uint32_t HINT_EXACT_OK = 1u, HINT_SINGLE_OK = 2u, HINT_L1 = 4u, HINT_R1 = 8u, HINT_PAR = 16u, HINT_ALT_OK = 32u;
constexpr uint32_t HINT_MULTI = 64u; // K_i has several labels, all (i, i+31) on one site: bits 8..15 of x are the set of their alleles
// HINT_TWO: K_i has 2..HINT_OWN_MAX labels, all (i, i+31), on TWO neighbouring sites s (x's site field) and s + 1, alleles
// 0..3: bits 8..11 of x are the set of s's alleles, bits 12..15 that of s + 1's (labels come in ascending variant id --
// IndexEntry::variant_id is a std::set -- so s's are first: the path's sites keep that order, path.cpp:105-129)
constexpr uint32_t HINT_TWO = 128u;
constexpr uint32_t HINT_OWN_MAX = 4u, HINT_MASK_BITS = 8u; // (express4's KS labels per k-mer; the allele numbers the byte holds)
constexpr uint32_t HINT_ALTIDX_SHIFT = 8u, HINT_SITE_SHIFT = 12u + 4u; // x: flags 0..7, allele numbers 8..15, site 16..31
constexpr uint32_t HINT_NO_SITE = 0xFFFFu;
constexpr uint32_t HINT_BACK_SHIFT = 8u, HINT_SNPOFF_SHIFT = 16u; // y
// y, with HINT_ALT_OK: HINT_SNP_GROUP = the keys of the site's alleles K_0 .. K_nv-1 (K_i with the base on the site replaced)
// are alone with their halves -- every K_a is the only indexed key with its 16 bases on the SNP's side, and the keys that
// share the 16 bases on the other side are exactly those nv keys; bits 22..23: the reference allele's base there (A C G T =
// 0..3), bits 24..26: nv.  With it a read k-mer over the SNP is judged against the allele it carries, whichever that is.
constexpr uint32_t HINT_SNP_GROUP = 1u << 21, HINT_REFB_SHIFT = 22u, HINT_NV_SHIFT = 24u;
// y bits 27..28 / 29..30 (HINT_FAR_LEFT / _RIGHT_SHIFT): how far the OTHER keys that share K_i's 16 first / last bases are from
// K_i, in substitutions: 0 = one of them within 2 (or the group was too crowded to look through), 1 = all at least 3 away, 2 = at
// least 4, 3 = at least 8 or there is none.  A read k-mer m substitutions from K_i inside one half can only meet keys of the
// other half's group, and those are within m (exact hit) / m + 1 (Hamming-1 neighbour) of K_i: HINT_FAR_NEED(m) rules them out.
constexpr uint32_t HINT_FAR_LEFT_SHIFT = 27u, HINT_FAR_RIGHT_SHIFT = 29u;
// y bit 31, HINT_NEAR_FREE (round 6): the half-key filters say "absent" for EVERY 16-mer that is one substitution away from the first
// or from the last 16 bases of the reference k-mer at the position (2 x 16 x 3 probes, made once when the tables are built).  A read
// k-mer whose half differs from the reference's there in exactly one unambiguous base would probe one of those 96: the answer is
// known, the probe -- a random 4-byte look into an 8 MB table, a sector of traffic per look and a quarter of what the
// position-hinted pass fetched -- is not made.  The pass decides exactly what it decided with the probe (the flag IS the probes'
// outcome); where a variant is in the filter (a SNP's other allele, a chance hit: 2-3 % of the positions) the flag is clear and
// the probe is made as before.
constexpr uint32_t HINT_NEAR_FREE = 1u << 31;
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr uint32_t hint_far_need(uint32_t m) // the code that proves "nobody within m + 1"
{
  return m <= 1 ? 1u : m == 2 ? 2u : m <= 6 ? 3u : 4u;
}
constexpr uint32_t HINT_TAIL_NODE = 2u; // tail_info.x: y is the position's reference node (set inside every reference node that has a site behind it)
constexpr uint32_t HINT_TAIL_OK = 1u, HINT_TAIL_NALL_SHIFT = 2u, HINT_TAIL_NEXT_SHIFT = 8u, HINT_TAIL_CODES_SHIFT = 16u; // tail_info.x
// HINT_EXACT_OK restates express4's seeding rule for an exact hit when the index is built; these are the limits of the
// rule's wide form (static_asserts in express4.inl): how many keys may share a half with the k-mer's key, how many labels
// its Hamming-1 neighbours may have together.  (The verdict is about the reference's lookups, not about which build of
// pass 1 runs behind pass 0: neighbours on the k-mer's own interval and site end where the chain ends, whatever their number.)
constexpr uint32_t HINT_HE_CAP = 16, HINT_NB_MAX = 16;

// hash of a 16-base half as two bit planes of its 2-bit codes (A0 C1 G2 T3): w0 = the low bits, w1 = the high bits, base j of
// Words of a half-key filter for n_keys indexed keys, as a power of two: 128 bits per key and side (round 6; 32 before).  A
// probe looks at four bits of ONE word, and with a key per word on average one probe in three hundred of a half that occurs
// nowhere met four set bits in a word that two or three keys share (cfg2: a third of what the position-hinted pass sent on,
// 7 k reads of 10 M), and one position in four failed HINT_NEAR_FREE for one of its 96 neighbours; with a key per four words it
// is one probe in five thousand and one position in fifty.  32 MB for both filters of a 1 Mb region's million keys.
// (GTX_FILTER_BITS_LOG2: A/B switch, 5 = the old size)
inline uint32_t hint_filter_log2_words(uint64_t n_keys)
{
  uint32_t fl = 5;
  while ((1ull << fl) < n_keys + 1 && fl < 28)
    ++fl;
  uint32_t extra = 2;
  if (char const * e = std::getenv("GTX_FILTER_BITS_LOG2"))
    extra = static_cast<uint32_t>(std::max(5, std::min(9, std::atoi(e))) - 5);
  return std::min<uint32_t>(fl + extra, 28u);
}

// the half at bit j (what a read in plane form yields with two shifts) -> (word, mask of up to four bits) of the blocked
// Bloom filter
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void hint_filter_hash(uint32_t w0, uint32_t w1, uint32_t & h, uint32_t & h2)
{
  // two 32-bit multiplies (the first form's two 64-bit multiplies were eight quarter-rate vector instructions per probe
  // slot, ten slots per read); the xor-shifts carry the well-mixed high bits down to the bits the mask is taken from
  h = w1 * 0x9E3779B1u;
  h ^= h >> 15;
  h += w0;
  h *= 0x85EBCA77u;
  h ^= h >> 16;
  h2 = h * 0x9E3779B1u; // (a second mix: the bits the masks are taken from)
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void hint_filter_slot_of(uint32_t h, uint32_t h2, uint32_t log2_words, uint32_t & word, uint32_t & mask)
{
  word = log2_words >= 32 ? h : h >> (32 - log2_words);
  // four bits of the word (32 bits per key and side: a word is about an eighth full, four bits make a
  // chance hit of a half that occurs nowhere rare; with two bits it was ~1 % of the probes: 8 % of what the pass declined at cfg2)
  mask = (1u << (h2 >> 27)) | (1u << ((h2 >> 22) & 31u)) | (1u << ((h2 >> 17) & 31u)) | (1u << ((h2 >> 12) & 31u));
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void hint_filter_slot(uint32_t w0, uint32_t w1, uint32_t log2_words, uint32_t & word, uint32_t & mask)
{
  uint32_t h, h2;
  hint_filter_hash(w0, w1, h, h2);
  hint_filter_slot_of(h, h2, log2_words, word, mask);
}

struct HostGraph
{
  std::vector<uint32_t> ref_order, ref_len, ref_dna, ref_nvar, ref_first_var;
  std::vector<uint32_t> var_order, var_len, var_dna, var_out_ref;
  std::vector<uint32_t> site_ref_reach, site_special_base, special_ref_reach, special_actual, pos_bucket, pos_info;
  std::vector<uint8_t> pos_back;
  std::vector<uint32_t> pos_node;
  std::vector<uint32_t> event_off; // [2*n_var+1] (empty when the graph has no events)
  std::vector<int64_t> event_val;
  std::vector<uint64_t> tri_off, allele_off, near_off;
  std::vector<uint32_t> near_last;
  std::string dna;   // characters (index construction)
  std::string codes; // the same arena as comparison codes (kernels)
  uint64_t total_tri = 0, total_allele = 0, total_near = 0;
  uint32_t n_hap = 0;
  uint32_t padding = 1000;
  bool is_sv_graph = false;
  // positions the pos_info / pos_back / pos_node tables cover (0: the arena is too large for them); the vectors themselves
  // are filled for host use only -- a device context makes the tables on its device (flatten_graph: host_positions)
  uint32_t pos_table_len = 0;

  GraphView view() const;
};

struct HostIndex
{
  // reference-order content: keys ascending, labels of one key in emission (bucket) order
  std::vector<uint64_t> keys;
  std::vector<uint32_t> key_off; // [n_keys+1]
  std::vector<gtx_label> labels;
  // device form
  std::vector<IndexSlot> slots;
  std::vector<DevLabel> dev_labels;
  uint32_t log2_cap = 0;
  // device tables (plane-form keys).  Half-key tables: hlist = all keys grouped by the 16 first bases followed by all
  // keys grouped by the 16 last bases; hslots maps (side, half) -> bucket
  std::vector<IndexSlot> hslots;
  std::vector<HalfEntry> hlist;
  uint32_t h_log2_cap = 0;
  // position-hinted pass (IndexView::refp ...)
  std::vector<uint32_t> refp, filt[2];
  std::vector<uint2_t> pos_flags, tail_info;
  uint32_t hint_first = 0, n_hint = 0, filt_log2 = 0;
  std::vector<HintWindow> win;
  std::vector<uint32_t> site_win;
  uint32_t win_base = 0;

  IndexView view(uint32_t max_index_labels, uint32_t half_bucket_cap) const; // over the host copies
};

// returns "" on success, else a description of what is wrong with the view
std::string flatten_graph(gtx_graph_view const & g, gtx_params const & par, HostGraph & out, bool host_positions = true);

// one indexed 32-mer occurrence as the enumeration emits it (emission order = the reference's bucket order)
struct Emit
{
  uint64_t key;
  gtx_label label;
};

// what the position-hinted pass needs of the GRAPH alone (no index): the linear reference as nibbles, per position the
// bases to the end / from the start of its reference node, the reference bit planes and the tail_info table
struct HintGraphTables
{
  std::vector<uint8_t> base, room, back;
  std::vector<uint32_t> refp;
  std::vector<uint2_t> tail_info;
  uint32_t hint_first = 0, n = 0; // n = 0: the graph has too many sites for the tables (no position-hinted pass)
};

// the in-node k-mers of reference node `node` from its 32nd position on (one per position), left to the device: `count` of
// them, behind `host_before` listed k-mers and `dev_before` k-mers of earlier runs in the sweep's order
struct EmitRun
{
  uint32_t node, host_before, dev_before, count;
};
void enumerate_kmers(HostGraph const & g, std::vector<Emit> & out);                 // index_graph's sweep (host threads)
void enumerate_kmers(HostGraph const & g, std::vector<Emit> & out, std::vector<EmitRun> * runs); // ... with the in-node runs left out
void hint_graph_tables(HostGraph const & g, HintGraphTables & out);
void hint_list_windows(HostGraph const & g, std::vector<HintWindow> & win, std::vector<uint32_t> & site_win);
void build_tables_host(HostGraph const & g, std::vector<Emit> const & em, HostIndex & out); // grouping, hash tables, hint tables
void build_index(HostGraph const & g, HostIndex & out);                             // both of the above

// Device tables are keyed by the k-mer in PLANE form: bit j of the low word = low bit of base j's 2-bit code, bit j of
// the high word = its high bit (base 0 = first base).  A wavefront gets both words straight from two ballots over the
// 32 bases, so no bit interleaving is needed on the device; the reference's key layout (type_conversions.cpp:75-87:
// first base in the top two bits) is kept for everything the host reports.
inline uint64_t plane_key(uint64_t key)
{
  // base j (0 = first base = top bit pair of `key`) -> bit j of the low-bit plane (low word) and of the high-bit plane
  auto compact_even_bits = [](uint64_t x) // bits 0,2,4,.. of x -> bits 0..31
  {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return static_cast<uint32_t>(x);
  };
  auto reverse32 = [](uint32_t v)
  {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
  };
  uint64_t const lo = reverse32(compact_even_bits(key)), hi = reverse32(compact_even_bits(key >> 1));
  return (hi << 32) | lo;
}

// Pass 1 of the alignment has a lean and a wide build (express4.inl).  The wide one runs ~45 % longer per read and
// finishes the reads whose k-mers lie over two or more variant sites, which the lean one hands to the general pass at
// ~20x the cost: it pays off once more than ~2 % of the indexed keys carry several labels (measured: a SNP every 100
// bases at regular distances -- no such key -- is 17 % faster with the lean build, a SNP every 25 bases -- every read
// has such a k-mer -- 1.8x faster with the wide one).
inline bool express4_prefers_wide(HostGraph const & g, uint64_t n_keys, uint64_t keys_with_several_labels)
{
  if (n_keys != 0 && keys_with_several_labels * 50 > n_keys)
    return true;
  // ... or once the walk at a read's end (26 characters of a 150 bp read) meets an indel site for more than ~8 % of the
  // reads: only the wide build walks over alleles of unequal length (an indel every 60 bases: 44 % of the tasks reach
  // the general pass with the lean build, 20 % with the wide one)
  uint64_t bases = 0, indel_sites = 0;
  for (std::size_t r = 0; r < g.ref_len.size(); ++r)
  {
    bases += g.ref_len[r];
    bool indel = false;
    for (uint32_t a = 0; a < g.ref_nvar[r]; ++a)
      indel = indel || g.var_len[g.ref_first_var[r] + a] != 1;
    indel_sites += indel ? 1u : 0u;
  }
  return indel_sites * 325 > bases;
}

inline bool express4_prefers_wide(HostGraph const & g, HostIndex const & ix)
{
  uint64_t several = 0;
  std::size_t const n = ix.keys.size();
  for (std::size_t k = 0; k < n; ++k)
    several += ix.key_off[k + 1] - ix.key_off[k] >= 2 ? 1u : 0u;
  return express4_prefers_wide(g, n, several);
}

#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t hash_key(uint64_t key, uint32_t log2_cap)
{
  return (key * 0x9E3779B97F4A7C15ull) >> (64 - log2_cap);
}

} // namespace gtx
