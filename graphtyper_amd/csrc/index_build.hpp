// index_build.hpp -- the per-key and per-position decisions behind the tables of the position-hinted pass
// (IndexView::pos_flags, filt; hinted.hpp), written once over plain arrays so that the host build (gtx_host.cpp, contexts
// without a device and the test emulation) and the device build (gtx_index_dev.hip: one thread per key / per position)
// run the same text.
#pragma once
#include "graph_dev.hpp"

#if defined(__HIPCC__)
#define GTX_HD __host__ __device__ inline
#else
#define GTX_HD inline
#endif

namespace gtx
{
// the finished index as plain arrays (host or device pointers)
struct HintKeys
{
  uint64_t const * keys;    // [n_keys] ascending, the reference's 2-bit layout (first base in the top bits)
  uint32_t const * key_off; // [n_keys + 1] labels of key k = labels[key_off[k] .. key_off[k + 1])
  DevLabel const * labels;  // in key order, bucket order inside a key
  uint32_t n_keys;
  // keys that share their 16 first bases are neighbours in `keys` (left group of k = keys [lbegin[k], lbegin[k] + lsize[k]));
  // rorder lists the key indices ordered by their 16 last bases (right group of k = rorder[rbegin[k] .. rbegin[k] + rsize[k]))
  uint32_t const * lbegin;
  uint32_t const * lsize;
  uint32_t const * rorder;
  uint32_t const * rbegin;
  uint32_t const * rsize;
  // the half-key filters, once they are made (hint_flags_at: HINT_NEAR_FREE; nullptr while they are not: the bit stays clear)
  uint32_t const * filt0 = nullptr;
  uint32_t const * filt1 = nullptr;
  uint32_t filt_log2 = 0;
};

// 16 bases (2 bits each, first base in the top bits) as the two planes the kernel hashes (hint_filter_slot): bit j of
// w0 / w1 = low / high bit of base j
GTX_HD void hint_half_planes(uint32_t half, uint32_t & w0, uint32_t & w1)
{
  w0 = w1 = 0;
  for (uint32_t j = 0; j < 16; ++j)
  {
    uint32_t const two = (half >> (30 - 2 * j)) & 3u;
    w0 |= (two & 1u) << j;
    w1 |= (two >> 1) << j;
  }
}

GTX_HD bool hint_distance1(uint64_t a, uint64_t b) // exactly one base differs
{
  uint64_t const x = a ^ b, bases = (x | (x >> 1)) & 0x5555555555555555ull;
  return bases != 0 && (bases & (bases - 1)) == 0;
}

// Hamming-1 neighbours of key k among the keys that share one of its halves: how many labels they have together (nb), and
// whether every one of those labels is k's own interval on k's own site (what express4's seeding rule asks of the
// neighbours of an exact hit).  A crowded group (more than 64 keys: low-complexity sequence) is not looked through: same = 0.
GTX_HD void hint_judge_key(HintKeys const & t, uint32_t k, uint32_t & nb, uint32_t & same, uint32_t & known)
{
  nb = 0;
  same = 1;
  // `known` (SLOT_NB_KNOWN): the key's own labels share one interval and all lie on sites; every neighbour label has that
  // interval and one of those sites (express4.inl: the neighbours of an exact hit)
  known = 1;
  uint32_t const own0 = t.key_off[k], own1 = t.key_off[k + 1];
  for (uint32_t i = own0; i < own1; ++i)
  {
    DevLabel const lb = t.labels[i];
    if (lb.site == INVALID || lb.start != t.labels[own0].start || lb.end != t.labels[own0].end)
      known = 0;
  }
  DevLabel const la = t.labels[t.key_off[k]];
  // (k's own labels: one, or up to HINT_OWN_MAX on one interval of one site -- the alleles of a merged site that share the k-mer)
  bool uniform = t.key_off[k + 1] - t.key_off[k] <= HINT_OWN_MAX;
  for (uint32_t i = t.key_off[k] + 1; i < t.key_off[k + 1] && uniform; ++i)
  {
    DevLabel const lb = t.labels[i];
    uniform = la.site != INVALID && lb.site == la.site && lb.start == la.start && lb.end == la.end;
  }
  auto look = [&](uint32_t b)
  {
    if (b == k || !hint_distance1(t.keys[k], t.keys[b]))
      return;
    nb += t.key_off[b + 1] - t.key_off[b];
    for (uint32_t i = t.key_off[b]; i < t.key_off[b + 1]; ++i)
    {
      DevLabel const lb = t.labels[i];
      if (!uniform || la.site == INVALID || lb.site != la.site || lb.start != la.start || lb.end != la.end)
        same = 0;
      bool among = false;
      for (uint32_t o = own0; o < own1 && !among && known; ++o)
        among = t.labels[o].site == lb.site;
      if (!among || lb.start != la.start || lb.end != la.end)
        known = 0;
    }
  };
  if (own1 - own0 > 64 || t.lsize[k] > 64 || t.rsize[k] > 64)
    known = 0;
  if (t.lsize[k] > 64)
    same = 0;
  else
    for (uint32_t i = 0; i < t.lsize[k]; ++i)
      look(t.lbegin[k] + i);
  if (t.rsize[k] > 64)
    same = 0;
  else
    for (uint32_t i = 0; i < t.rsize[k]; ++i)
      look(t.rorder[t.rbegin[k] + i]);
}

GTX_HD bool hint_find_key(HintKeys const & t, uint64_t key, uint32_t & k)
{
  uint32_t lo = 0, hi = t.n_keys;
  while (lo < hi)
  {
    uint32_t const mid = lo + (hi - lo) / 2;
    if (t.keys[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  k = lo;
  return lo < t.n_keys && t.keys[lo] == key;
}

// the verdict express4's seeding rule gives a read k-mer that equals indexed key k whose labels all lie on (order, order + 31)
// of one site (hint_own_labels below): its indexed neighbours have to be that same interval on that site
GTX_HD bool hint_neighbours_ok(HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint32_t k, bool & par)
{
  par = nb[k] != 0;
  return t.lsize[k] <= HINT_HE_CAP && t.rsize[k] <= HINT_HE_CAP && (nb[k] == 0 || ((nb_same[k] & 1u) && nb[k] <= HINT_NB_MAX));
}

// Key k's labels when there are several: all (order, order + 31) on one site, alleles below HINT_MASK_BITS -> their set
// (0 = not of that form).  These are the alleles of a merged site that share the k-mer (express4.inl: labels with equal
// ends are one path whose allele set is the union over the labels).
// (one label counts when it names another allele than the reference's: the k-mer of an allele window, or of a place whose
//  reference path the index sweep pruned)
GTX_HD uint32_t hint_own_labels(HintKeys const & t, uint32_t k, uint32_t exp_start, uint32_t exp_end, uint32_t & site)
{
  uint32_t const n = t.key_off[k + 1] - t.key_off[k];
  if (n < 1 || n > HINT_OWN_MAX)
    return 0;
  uint32_t mask = 0;
  site = t.labels[t.key_off[k]].site;
  for (uint32_t i = t.key_off[k]; i < t.key_off[k + 1]; ++i)
  {
    DevLabel const l = t.labels[i];
    if (l.start != exp_start || l.end != exp_end || l.site == INVALID || l.site != site || l.allele >= HINT_MASK_BITS)
      return 0;
    mask |= 1u << l.allele;
  }
  return (n == 1 && mask == 1u) ? 0u : mask; // (the reference allele alone: HINT_SINGLE_OK's case)
}

// ... or on two neighbouring sites s < s + 1 (HINT_TWO): the sets of their alleles in bits 0..3 / 4..7 (0 = not of that form)
GTX_HD uint32_t hint_own_labels_two(HintKeys const & t, uint32_t k, uint32_t exp_start, uint32_t exp_end, uint32_t & site)
{
  uint32_t const n = t.key_off[k + 1] - t.key_off[k];
  if (n < 2 || n > HINT_OWN_MAX)
    return 0;
  uint32_t lo = 0, hi = 0;
  site = t.labels[t.key_off[k]].site;
  if (site == INVALID)
    return 0;
  for (uint32_t i = t.key_off[k]; i < t.key_off[k + 1]; ++i)
  {
    DevLabel const l = t.labels[i];
    if (l.start != exp_start || l.end != exp_end || l.site == INVALID || (l.site != site && l.site != site + 1) || l.allele >= 4u)
      return 0;
    if (l.site == site)
      lo |= 1u << l.allele;
    else
      hi |= 1u << l.allele;
  }
  return hi == 0 ? 0u : lo | (hi << 4);
}

// express4's rule for the neighbours of an exact hit over several sites: every neighbour label is the k-mer's own interval
// on one of its own sites (hint_judge_key: `known`, bit 1 of nb_same)
GTX_HD bool hint_neighbours_known(HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint32_t k, bool & par)
{
  par = nb[k] != 0;
  return t.lsize[k] <= HINT_HE_CAP && t.rsize[k] <= HINT_HE_CAP && (nb[k] == 0 || ((nb_same[k] & 2u) && nb[k] <= HINT_NB_MAX));
}

// ... and with one label: it has to be (order, order + 31, site, allele)
GTX_HD bool hint_exact_verdict(HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint32_t k, uint32_t exp_start, uint32_t exp_end,
                                uint32_t want_site, uint32_t want_allele, bool & par)
{
  if (t.key_off[k + 1] - t.key_off[k] != 1)
    return false;
  DevLabel const l = t.labels[t.key_off[k]];
  if (l.start != exp_start || l.end != exp_end || l.site != want_site || (l.site != INVALID && l.allele != want_allele))
    return false;
  return hint_neighbours_ok(t, nb, nb_same, k, par);
}

GTX_HD uint32_t hint_two_bits(uint32_t nibble) // A=1 C=2 G=4 T=8 -> 0..3
{
  return nibble == 1 ? 0u : nibble == 2 ? 1u : nibble == 4 ? 2u : 3u;
}

GTX_HD uint32_t hint_acgt(uint8_t code) // graph comparison code -> nibble of A/C/G/T, 15 for anything else
{
  return (code == 1 || code == 2 || code == 4 || code == 8) ? code : 15u;
}

// ... -> the nibble the reference planes hold for that base: an IUPAC letter's own 4-bit code (the walks compare characters: a
// letter that is neither the read's nor N counts as a mismatch, graph_utils.hpp:7-69), 0 -- equal to no read base -- for
// anything that is no IUPAC letter
GTX_HD uint32_t hint_plane_code(uint8_t code)
{
  return (code >= 1 && code <= 15) ? code : 0u;
}

GTX_HD bool hint_is_acgt(uint32_t nibble)
{
  return nibble == 1 || nibble == 2 || nibble == 4 || nibble == 8;
}

// IndexView::pos_flags[p]: `base` = the linear reference as nibbles (15 = not ACGT), `room` / `back` = bases to the end /
// from the start of the position's reference node (capped at 255, 0 outside reference nodes), n = positions
// (exp_start, exp_end): where a label of the k-mer at p has to lie -- (first_order + p, + 31) on the linear reference, the
// window's path in an allele window (`linear` false: the offsets of sites under the k-mer are not plain differences of
// orders there, the SNP logic is left out)
GTX_HD uint2_t hint_flags_at(GraphView const & g, HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint8_t const * base,
                              uint8_t const * room, uint8_t const * back, uint32_t n, uint32_t p, uint32_t exp_start, uint32_t exp_end, bool linear)
{
  uint32_t x = 0, y = static_cast<uint32_t>(room[p]) | (static_cast<uint32_t>(back[p]) << HINT_BACK_SHIFT);
  uint32_t site = HINT_NO_SITE;
  uint64_t key = 0;
  bool valid = p + K <= n;
  for (uint32_t j = 0; j < K && valid; ++j)
  {
    uint32_t const c = base[p + j];
    valid = hint_is_acgt(c);
    key = (key << 2) | hint_two_bits(c);
  }
  // nobody a substitution away from one of the reference k-mer's halves is in the filters (HINT_NEAR_FREE, gtx_flat.hpp)
  if (valid && t.filt0 && t.filt1)
  {
    bool near_free = true;
    for (uint32_t side = 0; side < 2 && near_free; ++side)
    {
      uint32_t p0, p1; // the half as the two planes the kernel hashes: base j at bit j
      hint_half_planes(static_cast<uint32_t>(side == 0 ? key >> 32 : key), p0, p1);
      uint32_t const * const f = side == 0 ? t.filt0 : t.filt1;
      for (uint32_t j = 0; j < 16 && near_free; ++j)
        for (uint32_t d = 1; d < 4 && near_free; ++d) // (base j becomes another one: its two bits change by d)
        {
          uint32_t word, mask;
          hint_filter_slot(p0 ^ ((d & 1u) << j), p1 ^ ((d >> 1) << j), t.filt_log2, word, mask);
          near_free = (f[word] & mask) != mask;
        }
    }
    if (near_free)
      y |= HINT_NEAR_FREE;
  }
  uint32_t k = 0;
  bool const found = valid && hint_find_key(t, key, k);
  if (found)
  {
    // how far the other keys with K's halves are from K (HINT_FAR_*)
    for (uint32_t side = 0; side < 2; ++side)
    {
      uint32_t const n_group = side == 0 ? t.lsize[k] : t.rsize[k];
      uint32_t dmin = 64;
      if (n_group > 64)
        dmin = 0;
      else
        for (uint32_t i = 0; i < n_group; ++i)
        {
          uint32_t const b = side == 0 ? t.lbegin[k] + i : t.rorder[t.rbegin[k] + i];
          if (b == k)
            continue;
          uint64_t const d = t.keys[k] ^ t.keys[b];
          uint32_t const n_differ = static_cast<uint32_t>(__builtin_popcountll((d | (d >> 1)) & 0x5555555555555555ull));
          dmin = n_differ < dmin ? n_differ : dmin;
        }
      uint32_t const code = dmin >= 8 ? 3u : dmin >= 4 ? 2u : dmin >= 3 ? 1u : 0u;
      y |= code << (side == 0 ? HINT_FAR_LEFT_SHIFT : HINT_FAR_RIGHT_SHIFT);
    }
  }
  bool const one_alt_label = found && t.key_off[k + 1] - t.key_off[k] == 1 && t.labels[t.key_off[k]].site != INVALID && t.labels[t.key_off[k]].allele != 0;
  if (found && (t.key_off[k + 1] - t.key_off[k] > 1 || one_alt_label) && !g.is_sv_graph)
  {
    // the k-mer lies over a merged site and several of its alleles spell it
    uint32_t msite = INVALID;
    uint32_t const mask = hint_own_labels(t, k, exp_start, exp_end, msite);
    bool par = false;
    if (mask != 0 && msite < HINT_NO_SITE && hint_neighbours_ok(t, nb, nb_same, k, par))
    {
      site = msite;
      x |= HINT_EXACT_OK | HINT_MULTI | (par ? HINT_PAR : 0u) | (mask << HINT_ALTIDX_SHIFT);
    }
    else if (mask == 0)
    {
      // ... or the k-mer lies over two neighbouring sites
      uint32_t const two = hint_own_labels_two(t, k, exp_start, exp_end, msite);
      if (two != 0 && msite + 1 < HINT_NO_SITE && hint_neighbours_known(t, nb, nb_same, k, par))
      {
        site = msite;
        x |= HINT_EXACT_OK | HINT_TWO | (par ? HINT_PAR : 0u) | (two << HINT_ALTIDX_SHIFT);
      }
    }
  }
  if (found && t.key_off[k + 1] - t.key_off[k] == 1)
  {
    DevLabel const l = t.labels[t.key_off[k]];
    uint32_t const order = exp_start;
    if (l.start == exp_start && l.end == exp_end && (l.site == INVALID || l.allele == 0) && !(l.site != INVALID && g.is_sv_graph))
    {
      site = l.site == INVALID ? HINT_NO_SITE : l.site;
      x |= HINT_SINGLE_OK;
      if (t.lsize[k] == 1)
        x |= HINT_L1;
      if (t.rsize[k] == 1)
        x |= HINT_R1;
      bool par = false;
      if (hint_exact_verdict(t, nb, nb_same, k, exp_start, exp_end, l.site, 0, par))
        x |= HINT_EXACT_OK | (par ? HINT_PAR : 0u);
      // the other alleles of a SNP under the k-mer
      if (l.site != INVALID && linear)
      {
        uint32_t const fv = g.ref_first_var[l.site], nv = g.ref_nvar[l.site];
        bool snp = nv >= 2 && nv <= 4 && g.var_order[fv] >= order && g.var_order[fv] <= order + K - 1;
        for (uint32_t a = 0; a < nv && snp; ++a)
          snp = g.var_len[fv + a] == 1 && hint_acgt(static_cast<uint8_t>(g.dna[g.var_dna[fv + a]])) != 15;
        if (snp)
        {
          uint32_t const off = g.var_order[fv] - order; // base of the k-mer that lies on the site
          uint32_t idx_of = 0;
          // the allele keys alone with their halves (HINT_SNP_GROUP): on the SNP's side every key by itself, on the other
          // side the nv of them together (they share it by construction: a group of nv keys holds nobody else)
          bool const snp_left = off < K / 2;
          bool group = (snp_left ? t.lsize[k] : t.rsize[k]) == 1 && (snp_left ? t.rsize[k] : t.lsize[k]) == nv;
          for (uint32_t a = 1; a < nv && snp; ++a)
          {
            uint32_t const two = hint_two_bits(hint_acgt(static_cast<uint8_t>(g.dna[g.var_dna[fv + a]])));
            uint64_t const alt = (key & ~(3ull << (2 * (K - 1 - off)))) | (static_cast<uint64_t>(two) << (2 * (K - 1 - off)));
            uint32_t ka = 0;
            bool pa = false;
            snp = alt != key && ((idx_of >> (2 * two)) & 3u) == 0 && hint_find_key(t, alt, ka) && hint_exact_verdict(t, nb, nb_same, ka, exp_start, exp_end, l.site, a, pa);
            idx_of |= a << (2 * two);
            group = group && snp && (snp_left ? t.lsize[ka] : t.rsize[ka]) == 1;
          }
          if (snp)
          {
            x |= HINT_ALT_OK | (idx_of << HINT_ALTIDX_SHIFT);
            y |= off << HINT_SNPOFF_SHIFT;
            uint32_t const refb = static_cast<uint32_t>(key >> (2 * (K - 1 - off))) & 3u;
            y |= (group ? HINT_SNP_GROUP : 0u) | (refb << HINT_REFB_SHIFT) | (nv << HINT_NV_SHIFT);
          }
        }
      }
    }
  }
  return uint2_t{x | (site << HINT_SITE_SHIFT), y};
}

GTX_HD uint2_t hint_position_flags(GraphView const & g, HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint8_t const * base,
                                    uint8_t const * room, uint8_t const * back, uint32_t n, uint32_t p)
{
  return hint_flags_at(g, t, nb, nb_same, base, room, back, n, p, g.first_order + p, g.first_order + p + K - 1, true);
}

// ---- allele windows (IndexView::win) ----
// Which sites get windows for their alternative alleles: those whose reads the linear reference's flags cannot settle -- an
// allele of another length than one base, more than four alleles, or another site within a k-mer's reach (a SNP standing
// alone is settled by HINT_SNP_GROUP)
GTX_HD bool hint_site_wants_windows(GraphView const & g, uint32_t r)
{
  if (g.is_sv_graph || r + 1 >= g.n_ref)
    return false;
  uint32_t const nv = g.ref_nvar[r], fv = g.ref_first_var[r];
  if (nv < 2 || nv > HINT_MASK_BITS || g.var_len[fv] == 0 || g.var_len[fv] > HINT_WIN_ALLELE_MAX)
    return false;
  bool plain = nv <= 4;
  for (uint32_t a = 0; a < nv; ++a)
    plain = plain && g.var_len[fv + a] == 1;
  if (!plain)
    return true;
  bool const near_prev = r > 0 && g.ref_nvar[r - 1] != 0 && g.ref_len[r] < K - 1;
  bool const near_next = r + 2 < g.n_ref && g.ref_nvar[r + 1] != 0 && g.ref_len[r + 1] < K - 1;
  return near_prev || near_next;
}

GTX_HD bool hint_allele_gets_window(GraphView const & g, uint32_t r, uint32_t a)
{
  uint32_t const len = g.var_len[g.ref_first_var[r] + a];
  return a != 0 && len != 0 && len <= HINT_WIN_ALLELE_MAX;
}

// one position of a window: base / room / back / tail entry (those of the linear reference's position it copies; inside the
// allele: the allele's base, no node).  The node in front of the window's own site loses HINT_TAIL_OK: what hint_compare
// counted on the site's position is the window's allele there, not the reference allele site_choice() assumes.
GTX_HD void hint_window_cell(GraphView const & g, HintWindow const & w, uint32_t local, uint8_t const * mbase, uint8_t const * mroom,
                             uint8_t const * mback, uint2_t const * mtail, uint32_t n_main, uint8_t & base, uint8_t & room, uint8_t & back, uint2_t & tail)
{
  base = 15;
  room = back = 0;
  tail = uint2_t{0, 0};
  uint32_t order = 0;
  if (local < HINT_WIN_BEFORE)
  {
    if (w.site_order < g.first_order + (HINT_WIN_BEFORE - local))
      return;
    order = w.site_order - (HINT_WIN_BEFORE - local);
  }
  else
  {
    uint32_t const k = local - HINT_WIN_BEFORE;
    if (k < w.len_a)
    {
      base = static_cast<uint8_t>(hint_plane_code(static_cast<uint8_t>(g.dna[g.var_dna[g.ref_first_var[w.site] + w.allele] + k])));
      return;
    }
    if (k - w.len_a >= HINT_WIN_BEFORE)
      return;
    order = w.site_order + w.len_0 + (k - w.len_a);
  }
  uint32_t const m = order - g.first_order;
  if (m >= n_main)
    return;
  base = mbase[m];
  room = mroom[m];
  back = mback[m];
  tail = mtail[m];
  if ((tail.x & HINT_TAIL_NODE) != 0 && tail.y == w.site)
    tail.x = HINT_TAIL_NODE;
}

// ... and its flags: a k-mer that does not reach into the allele is a k-mer of the linear reference (that position's flags), one
// that does is judged against the window's path
GTX_HD uint2_t hint_window_flags(GraphView const & g, HintKeys const & t, uint32_t const * nb, uint8_t const * nb_same, uint8_t const * base,
                                  uint8_t const * room, uint8_t const * back, uint32_t n_total, uint2_t const * main_flags, uint32_t n_main,
                                  HintWindow const & w, uint32_t local, uint32_t p)
{
  uint2_t const none{HINT_NO_SITE << HINT_SITE_SHIFT, static_cast<uint32_t>(room[p]) | (static_cast<uint32_t>(back[p]) << HINT_BACK_SHIFT)};
  if (local + K > HINT_WIN_STRIDE)
    return none;
  bool const reaches = local + K > HINT_WIN_BEFORE && local < HINT_WIN_BEFORE + w.len_a;
  if (!reaches)
  {
    uint32_t order = 0;
    if (local < HINT_WIN_BEFORE)
    {
      if (w.site_order < g.first_order + (HINT_WIN_BEFORE - local))
        return none;
      order = w.site_order - (HINT_WIN_BEFORE - local);
    }
    else
      order = w.site_order + w.len_0 + (local - HINT_WIN_BEFORE - w.len_a);
    uint32_t const m = order - g.first_order;
    return m < n_main ? main_flags[m] : none;
  }
  return hint_flags_at(g, t, nb, nb_same, base, room, back, n_total, p, hint_win_order(g, w, local), hint_win_order(g, w, local + K - 1), false);
}

} // namespace gtx
