// align_core.inl -- the table-size dependent part of the alignment kernel.  Included inside a namespace that defines
// `struct AlignCfg` (see align_core.hpp: once with the LDS-sized tables of the main pass, once with the large tables of
// the second pass for reads that overflowed).  No include guard on purpose.

struct Here {}; // tag of this instantiation's namespace: keeps argument-dependent lookup from finding the other instantiation

struct PVar
{
  uint32_t site;
  uint32_t m[AlignCfg::MW]; // allele set (Path::nums[i]) as a bit mask of 32 * MW alleles: two words in the LDS and HBM passes,
                            // MAX_NUMBER_OF_HAPLOTYPES bits (constants.hpp.in:23) in the pass for graphs with wider sites
};
constexpr uint32_t PVAR_WORDS = 1 + AlignCfg::MW;

struct DPath // gyper::Path (include/graphtyper/typer/path.hpp:18-79)
{
  uint32_t start, end;
  uint16_t rs, re; // read_start_index, read_end_index
  uint16_t mism, nvar;
  PVar v[AlignCfg::MAXV];
};
constexpr uint32_t DPATH_WORDS = sizeof(DPath) / 4;

struct Loc // gyper::Location (include/graphtyper/graph/location.hpp)
{
  uint32_t type; // 0 = 'U', 1 = 'R', 2 = 'V'
  uint32_t node, order, offset;
};

struct Cand // one element of var_and_refs / var_ids / end_pos in Graph::get_labels_forward (graph.cpp:1192-1196)
{
  uint32_t len;  // var_and_refs[j].size()
  uint32_t mism; // mismatches of its first min(len, L) characters against the sub-read; max+1 once it is dead
  uint32_t pos;  // end_pos[j] (forward) or start_pos[j] (backward)
  uint32_t nids;
  uint32_t ids[AlignCfg::MAXIDS];
};
constexpr uint32_t CAND_WORDS = sizeof(Cand) / 4;

// A table of the workspace: an array of the configuration's size, or -- in the exact pass (AlignCfg::DYN, align_core.hpp) --
// a pointer into the task's slab of HBM whose capacity is a run-time value of the workspace (cap_* below).  Indexing,
// decay to a pointer and pointer arithmetic read the same on both.
template <class T, uint32_t N>
using TableOf = typename std::conditional<AlignCfg::DYN, T *, T[N]>::type;

// ... and the two tables of paths of the exact pass have a run-time PITCH as well: a path there has room for cap_v variant
// sites (24 in the launch that gives every task a part of the slab, AlignCfg::MAXV -- the proven bound -- in the launch with
// the whole slab), which makes it a tenth of the full struct and the part's capacity ten times what it would be.
struct PathTable
{
  uint8_t * base;
  uint32_t pitch;
  GTX_DEV DPath & operator[](uint32_t i) const { return *reinterpret_cast<DPath *>(base + static_cast<uint64_t>(i) * pitch); }
};
template <uint32_t N>
using PathsOf = typename std::conditional<AlignCfg::DYN, PathTable, DPath[N]>::type;

struct WalkBuffers // alive only during walk_read_starts / walk_read_ends
{
  PathsOf<AlignCfg::MAXPP> pp;
  TableOf<Cand, AlignCfg::CAND_CAP> cand;
  Loc locs[AlignCfg::LOC_CAP];
  TableOf<DevLabel, AlignCfg::WL_CAP> dfs_out; // labels of the current iterative_dfs call
};

// keybuf and the walk buffers are never alive at the same time: one storage for both -- unless the walk buffers are
// pointers (exact pass), which the keys must not overwrite
union WorkUnion
{
  uint64_t keybuf[AlignCfg::KEY_CAP]; // keys of a multi-key list, then (offset | count << 32) of every probed key
  WalkBuffers w;                      // (pp is also used while seeding, never at the same time as keybuf)
};
struct WorkBoth
{
  uint64_t keybuf[AlignCfg::KEY_CAP];
  WalkBuffers w;
};

// run-time table sizes and the side tables of the exact pass (empty in the passes with fixed tables)
struct DynTables
{
  uint32_t cap_lbl, cap_p, cap_cand, cap_wl; // entries of lbl; of paths and pp; of cand; of wl and of dfs_out
  uint32_t cap_v;                            // variant sites a path of the two path tables has room for
  uint32_t * pp_start, * pp_end;             // start / end of every pp entry, densely: what the searches of the chaining read
  uint64_t * bits_p, * bits_pp;              // one bit per path / per pp entry (the drop and matched sets of the filters and the chaining)
};
struct NoDynTables
{
};
// The passes with large fixed tables in HBM (AlignCfg::MAXPP > 64) keep the same dense start / end tables of pp, behind
// pointers: the kernel points them at LDS (a few KB), where the 64-entries-per-step searches of the chaining cost a hundred
// cycles instead of a round trip to HBM per step (round 4: a task with 512 paths took 50 ms in the HBM-table pass).
struct PpKeyTables
{
  uint32_t * pp_start, * pp_end;
  uint64_t * bits_pp; // one bit per pp entry (the bulk chaining's matched set)
};
constexpr bool DENSE_PP_KEYS = AlignCfg::DYN || AlignCfg::MAXPP > 64;

struct AlignWorkspace : std::conditional<AlignCfg::DYN, DynTables, std::conditional<(AlignCfg::MAXPP > 64), PpKeyTables, NoDynTables>::type>::type // lives in LDS (HBM in the passes behind the general one), one per wavefront
{
  uint8_t rd[AlignCfg::MAX_READ]; // read as 4-bit IUPAC codes, orientation applied
  TableOf<DevLabel, AlignCfg::LBL_CAP> lbl;
  PathsOf<AlignCfg::MAXP> paths;
  DPath orig, np; // Path temporaries of add_next/prev_kmer_labels
  typename std::conditional<AlignCfg::DYN, WorkBoth, WorkUnion>::type u;
  TableOf<DevLabel, AlignCfg::WL_CAP> wl; // best label lists of a walk (must survive the add_*_kmer_labels calls)
  uint32_t wl_off[AlignCfg::WLISTS + 1];
  uint32_t wl_idx[AlignCfg::WLISTS];
  // per k-mer exact-probe results
  uint64_t key0[AlignCfg::MAX_KMERS];
  uint32_t nkeys0[AlignCfg::MAX_KMERS];
  uint32_t off0[AlignCfg::MAX_KMERS];
  uint32_t cnt0[AlignCfg::MAX_KMERS];
  // lookups of the first KC k-mers, issued together so that their memory latencies overlap
  uint32_t hoff[AlignCfg::KC][2], hcnt[AlignCfg::KC][2];
  HalfEntry he[AlignCfg::KC][2][AlignCfg::HE_CAP];
  DevLabel xl[AlignCfg::KC][AlignCfg::XL_CAP];
  uint32_t fs_start[AlignCfg::KC], fs_end[AlignCfg::KC]; // the one label of every k-mer (fast seeding)
  uint32_t fs_site, fs_allele;                           // ... and the one variant among them
  uint32_t aoff[AlignCfg::MAX_KMERS][4], acnt[AlignCfg::MAX_KMERS][4]; // exact slots of the keys of a k-mer with one ambiguous base
  uint32_t n_paths, longest, status, n_lbl, n_keys, read_len, n_wl, n_wlists;
#ifdef GTX_PROF
  unsigned long long prof_acc[16]; // phase cycle sums of this wave (profiling build)
#endif
};

// What the first pass (gtx_align_express_kernel) needs: the read, its keys and staged lookups, and room for the single
// path it can produce.  Field names are those of AlignWorkspace: the seeding code is written once over either.
struct SeedWorkspace
{
  uint8_t rd[AlignCfg::MAX_READ];
  DPath paths[1];
  uint64_t key0[AlignCfg::MAX_KMERS];
  uint32_t nkeys0[AlignCfg::MAX_KMERS];
  uint32_t off0[AlignCfg::MAX_KMERS];
  uint32_t cnt0[AlignCfg::MAX_KMERS];
  uint32_t hoff[AlignCfg::KC][2], hcnt[AlignCfg::KC][2];
  HalfEntry he[AlignCfg::KC][2][AlignCfg::HE_CAP];
  DevLabel xl[AlignCfg::KC][AlignCfg::XL_CAP];
  uint32_t fs_start[AlignCfg::KC], fs_end[AlignCfg::KC];
  uint32_t fs_site, fs_allele;
  uint32_t aoff[AlignCfg::MAX_KMERS][4], acnt[AlignCfg::MAX_KMERS][4];
  uint32_t read_len;
#ifdef GTX_PROF
  unsigned long long prof_acc[16];
#endif
};

// ---- exact pass: cutting a slab of HBM into the tables of one task (AlignCfg::DYN; align_core.hpp: namespace exact) ----
// The slab starts with the workspace itself; behind it, for a path capacity P: paths and pp (P entries each), lbl
// (P + 128: a single exact key has no limit, a multi-key list stops at max_index_labels), wl and dfs_out (4 P + 4096
// labels each), the walk candidates (cand_cap entries: the reference stops branching at 128 live sequences, one more round
// over a site of n alleles makes at most 128 n of them -- exact_cand_cap of the graph's widest site), the dense start / end
// tables of pp and two bit sets.  exact_path_capacity is the largest P that fits (0: the slab cannot even hold the fixed part).
GTX_HDI uint32_t exact_cand_cap(uint32_t widest_site)
{
  return 128u * (widest_site < 2u ? 2u : widest_site) + 64u;
}

GTX_HDI uint64_t exact_fixed_bytes(uint32_t cand_cap)
{
  return ((sizeof(AlignWorkspace) + 255u) & ~static_cast<uint64_t>(255u)) + static_cast<uint64_t>(cand_cap) * sizeof(Cand) +
         (128u + 2u * 4096u) * sizeof(DevLabel) + 16u * 256u; // (+ alignment slack of the ten tables)
}

// bytes of one path with room for cap_v variant sites (the head of DPath + cap_v entries, 16-byte aligned)
GTX_HDI uint32_t exact_path_pitch(uint32_t cap_v)
{
  return (16u + cap_v * static_cast<uint32_t>(sizeof(PVar)) + 15u) & ~15u;
}

GTX_HDI uint64_t exact_bytes_per_path(uint32_t cap_v)
{
  return 2u * exact_path_pitch(cap_v) + 9u * sizeof(DevLabel) + 2u * sizeof(uint32_t) + 1u; // (+ two bits, rounded up)
}

GTX_HDI uint32_t exact_path_capacity(uint64_t slab_bytes, uint32_t cand_cap, uint32_t cap_v)
{
  uint64_t const fixed = exact_fixed_bytes(cand_cap) + 2u * 64u;
  if (slab_bytes <= fixed)
    return 0;
  uint64_t const p = (slab_bytes - fixed) / exact_bytes_per_path(cap_v);
  return p > 0x7FFFFFFFull ? 0x7FFFFFFFu : static_cast<uint32_t>(p);
}

// Points the tables of the workspace at the head of `slab` into it; cap_v: the variant sites a path gets room for (at most
// AlignCfg::MAXV).  Returns false when the slab is too small for a single path.
template <class W, class WS>
GTX_DEV bool exact_setup(WS * ws_at_slab_head, uint64_t slab_bytes, uint32_t cand_cap, uint32_t cap_v)
{
  if constexpr (AlignCfg::DYN)
  {
    WS & ws = *ws_at_slab_head;
    cap_v = cap_v < 1u ? 1u : cap_v > AlignCfg::MAXV ? AlignCfg::MAXV : cap_v;
    uint32_t const P = exact_path_capacity(slab_bytes, cand_cap, cap_v);
    if (P == 0)
      return false;
    GTX_LEAD
    {
      uint8_t * at = reinterpret_cast<uint8_t *>(ws_at_slab_head) + ((sizeof(WS) + 255u) & ~static_cast<uint64_t>(255u));
      auto take = [&](uint64_t bytes)
      {
        uint8_t * r = at;
        at += (bytes + 255u) & ~static_cast<uint64_t>(255u);
        return r;
      };
      uint32_t const pitch = exact_path_pitch(cap_v);
      ws.cap_p = P;
      ws.cap_v = cap_v;
      ws.cap_lbl = P + 128u;
      ws.cap_wl = 4u * P + 4096u;
      ws.cap_cand = cand_cap;
      ws.paths = PathTable{take(static_cast<uint64_t>(P) * pitch), pitch};
      ws.u.w.pp = PathTable{take(static_cast<uint64_t>(P) * pitch), pitch};
      ws.lbl = reinterpret_cast<DevLabel *>(take(static_cast<uint64_t>(ws.cap_lbl) * sizeof(DevLabel)));
      ws.wl = reinterpret_cast<DevLabel *>(take(static_cast<uint64_t>(ws.cap_wl) * sizeof(DevLabel)));
      ws.u.w.dfs_out = reinterpret_cast<DevLabel *>(take(static_cast<uint64_t>(ws.cap_wl) * sizeof(DevLabel)));
      ws.u.w.cand = reinterpret_cast<Cand *>(take(static_cast<uint64_t>(cand_cap) * sizeof(Cand)));
      ws.pp_start = reinterpret_cast<uint32_t *>(take(static_cast<uint64_t>(P) * sizeof(uint32_t)));
      ws.pp_end = reinterpret_cast<uint32_t *>(take(static_cast<uint64_t>(P) * sizeof(uint32_t)));
      ws.bits_p = reinterpret_cast<uint64_t *>(take((static_cast<uint64_t>(P) + 63u) / 64u * 8u));
      ws.bits_pp = reinterpret_cast<uint64_t *>(take((static_cast<uint64_t>(P) + 63u) / 64u * 8u));
    }
    W::lds_sync();
    return true;
  }
  else
    return true;
}

// table capacities: the configuration's constants, or what the task's slab holds (exact pass)
template <class WS>
GTX_DEV uint32_t cap_lbl(WS const & ws)
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_lbl;
  else
    return AlignCfg::LBL_CAP;
}
template <class WS>
GTX_DEV uint32_t cap_paths(WS const & ws)
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_p;
  else
    return AlignCfg::MAXP;
}
template <class WS>
GTX_DEV uint32_t cap_pp(WS const & ws)
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_p;
  else
    return AlignCfg::MAXPP;
}
template <class WS>
GTX_DEV uint32_t cap_sites(WS const & ws) // variant sites of a path
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_v;
  else
    return AlignCfg::MAXV;
}
template <class WS>
GTX_DEV uint32_t cap_cand(WS const & ws)
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_cand;
  else
    return AlignCfg::CAND_CAP;
}
template <class WS>
GTX_DEV uint32_t cap_wl(WS const & ws)
{
  if constexpr (AlignCfg::DYN)
    return ws.cap_wl;
  else
    return AlignCfg::WL_CAP;
}

// the dense start (chaining forwards) or end (backwards) table of the pp entries (exact pass)
template <class WS>
GTX_DEV uint32_t const * pp_keys(WS const & ws, bool ends)
{
  if constexpr (DENSE_PP_KEYS)
    return ends ? ws.pp_end : ws.pp_start;
  else
    return nullptr;
}

// A set of path (or pp) numbers: registers in the passes with fixed tables (BitSet, graph_dev.hpp), words of the task's
// slab in the exact pass.  Wave-uniform like everything else here: every lane sets the same bit and reads the same word.
struct MemBits
{
  uint64_t * w;
};
template <class W>
GTX_DEV void bits_set(MemBits & b, uint32_t i)
{
  GTX_LEAD b.w[i >> 6] |= 1ull << (i & 63u);
  W::lds_sync();
}
template <class W>
GTX_DEV bool bits_get(MemBits const & b, uint32_t i)
{
  return (GTX_U(b.w[i >> 6]) >> (i & 63u)) & 1ull;
}
// 64 members at once: word `k` of the set
template <class W>
GTX_DEV uint64_t bits_word(MemBits const & b, uint32_t k)
{
  return GTX_U(b.w[k]);
}
// ... and word `k` of the set becomes `mask`
template <class W>
GTX_DEV void bits_set_word(MemBits & b, uint32_t k, uint64_t mask)
{
  GTX_LEAD b.w[k] = mask;
  W::lds_sync();
}

// an empty set over n elements
template <class W>
GTX_DEV MemBits bits_clear(uint64_t * words, uint32_t n)
{
  uint32_t const nw = (n + 63u) / 64u;
  for (uint32_t b = 0; b < nw; b += 64)
    W::lanes([&](uint32_t l) {
      if (b + l < nw)
        words[b + l] = 0;
    });
  W::lds_sync();
  return MemBits{words};
}
template <class W, class WS>
GTX_DEV auto new_path_set(WS & ws, uint32_t n)
{
  if constexpr (AlignCfg::DYN)
    return bits_clear<W>(ws.bits_p, n);
  else
    return BitSet<AlignCfg::MAXP>();
}
template <class W, class WS>
GTX_DEV auto new_pp_set(WS & ws, uint32_t n)
{
  if constexpr (AlignCfg::DYN)
    return bits_clear<W>(ws.bits_pp, n);
  else
    return BitSet<AlignCfg::MAXPP>();
}

// allele sets.  Readers apply GTX_U per word (the tables are wave-uniform), writers run on the leader lane.
GTX_DEV bool pv_fits(uint32_t allele) // an allele beyond this pass' masks: the task overflows to the pass that holds it
{
  return allele < 32u * AlignCfg::MW;
}

template <class W>
GTX_DEV bool pv_has(PVar const & v, uint32_t allele)
{
  return allele < 32u * AlignCfg::MW && ((GTX_U(v.m[allele >> 5]) >> (allele & 31u)) & 1u);
}

GTX_DEV void pv_set_single(PVar & v, uint32_t allele)
{
  for (uint32_t w = 0; w < AlignCfg::MW; ++w)
    v.m[w] = 0;
  v.m[allele >> 5] = 1u << (allele & 31u);
}

GTX_DEV void pv_add(PVar & v, uint32_t allele)
{
  v.m[allele >> 5] |= 1u << (allele & 31u);
}

GTX_DEV void pv_clear(PVar & v)
{
  for (uint32_t w = 0; w < AlignCfg::MW; ++w)
    v.m[w] = 0;
}

template <class W>
GTX_DEV bool pv_meet_any(PVar const & a, PVar const & b)
{
  uint32_t any = 0;
  for (uint32_t w = 0; w < AlignCfg::MW; ++w)
    any |= GTX_U(a.m[w]) & GTX_U(b.m[w]);
  return any != 0;
}

GTX_DEV void pv_meet(PVar & a, PVar const & b)
{
  for (uint32_t w = 0; w < AlignCfg::MW; ++w)
    a.m[w] &= b.m[w];
}

// copy of a path: only the part in use (a path of the wide-site pass is 13 KB, nearly all of it unused mask words)
template <class W>
GTX_DEV void copy_path(DPath & dst, DPath const & src)
{
  if constexpr (DPATH_WORDS <= 64)
    copy_entry<W>(dst, src);
  else
  {
    if (&dst == &src)
      return;
    uint32_t const words = 4 + PVAR_WORDS * GTX_U(static_cast<uint32_t>(src.nvar));
    uint32_t * d = reinterpret_cast<uint32_t *>(&dst);
    uint32_t const * s = reinterpret_cast<uint32_t const *>(&src);
    W::lanes([&](uint32_t l) {
      for (uint32_t w = l; w < words; w += 64)
        d[w] = s[w];
    });
    W::lds_sync();
  }
}

GTX_DEV uint32_t path_size(DPath const & p)
{
  return static_cast<uint32_t>(p.re) - static_cast<uint32_t>(p.rs) + 1u;
}

template <class W>
GTX_DEV uint32_t upath_size(DPath const & p) // wave-uniform
{
  uint32_t const w = GTX_U(reinterpret_cast<uint32_t const *>(&p)[2]); // rs | re << 16
  return (w >> 16) - (w & 0xFFFFu) + 1u;
}

// Graph::get_locations_of_a_position (graph.cpp:1154-1185 -> 931-1029).  The reference scans reference nodes
// backwards and looks every variant node up in path.var_order; here the (few) sites of the path are visited in
// descending order instead, which yields the same locations in the same order.
template <class W>
GTX_DEV uint32_t get_locations(GraphView const & g, uint32_t pos, DPath const & path, Loc * locs, uint32_t cap, uint32_t & status)
{
  pos = GTX_U(pos);
  bool const special = g_is_special(g, pos);
  if (special)
    pos = GTX_U(g.special_actual[pos - SPECIAL_START]);
  uint32_t n = 0;
  if (pos < g.first_order)
    return 0;
  if (g.n_ref == 1)
  {
    GTX_LEAD locs[0] = Loc{1, 0, g.ref_order[0], pos - g.ref_order[0]};
    return 1;
  }
  int32_t rr = static_cast<int32_t>(g_ref_node_at<W>(g, pos));
  {
    uint32_t const ro = GTX_U(g.ref_order[rr]), rl = GTX_U(g.ref_len[rr]);
    if (pos < ro + rl)
    {
      if (!special)
      {
        GTX_LEAD locs[0] = Loc{1, static_cast<uint32_t>(rr), ro, pos - ro};
        return 1;
      }
      --rr;
    }
  }
  // sites rr' <= rr with reach(rr') + PADDING > pos, descending; only sites the path carries can contribute
  bool const path_empty = GTX_U(path.start) == GTX_U(path.end);
  uint32_t const nvar = GTX_U(static_cast<uint32_t>(path.nvar));
  int32_t bound = rr + 1;
  for (;;)
  {
    int32_t best = -1;
    uint32_t best_j = 0;
    for (uint32_t j = 0; j < nvar; ++j)
    {
      int32_t const s = static_cast<int32_t>(GTX_U(path.v[j].site));
      if (s < bound && s > best)
      {
        best = s;
        best_j = j;
      }
    }
    if (best < 0)
      break;
    bound = best;
    uint32_t const site = static_cast<uint32_t>(best);
    int64_t const reach = static_cast<int64_t>(GTX_U(g.ref_order[site])) + GTX_U(g.ref_len[site]) - 1;
    if (!(reach + static_cast<int64_t>(g.padding) > static_cast<int64_t>(pos)))
      break; // the reference stops its backward scan here; lower sites reach even less far
    uint32_t const fv = GTX_U(g.ref_first_var[site]), nv = GTX_U(g.ref_nvar[site]);
    // which alleles hold the position: lane i asks for allele i (one round trip for the site instead of two dependent
    // fetches per allele), the hits are then taken in allele order
    unsigned long long hits = 0;
    typename W::template PerLane<uint32_t> vo_l;
    if (nv <= 64)
    {
      typename W::template PerLane<bool> hit_l;
      W::lanes([&](uint32_t l) {
        bool h = false;
        uint32_t vo = 0;
        if (l < nv)
        {
          vo = g.var_order[fv + l];
          h = pos >= vo && pos <= vo + g.var_len[fv + l] - 1 &&
              (path_empty || (l < 32u * AlignCfg::MW && ((path.v[best_j].m[l >> 5] >> (l & 31u)) & 1u)));
        }
        hit_l[l] = h;
        vo_l[l] = vo;
      });
      hits = W::ballot(hit_l);
    }
    for (uint32_t i = 0; i < nv; ++i)
    {
      if (nv <= 64)
      {
        if (hits == 0)
          break;
        i = static_cast<uint32_t>(__builtin_ctzll(hits));
        hits &= hits - 1;
      }
      uint32_t const v = fv + i;
      uint32_t const vo = nv <= 64 ? W::from_lane(vo_l, i) : GTX_U(g.var_order[v]);
      if (nv <= 64 || (pos >= vo && pos <= vo + GTX_U(g.var_len[v]) - 1))
        if (nv <= 64 || path_empty || pv_has<W>(path.v[best_j], i))
        {
          if (n >= cap)
          {
            // beyond the table: with room for MAX_NUM_LOCATIONS_PER_PATH the caller only needs the count (it skips a path
            // with more locations, genotype_paths.cpp:508, 585: a site of a thousand alleles has that many), else the
            // task is for the pass with the larger table
            if (cap < MAX_NUM_LOCATIONS_PER_PATH)
            {
              status |= GTX_ST_DFS_OVERFLOW;
              return n;
            }
            ++n;
            continue;
          }
          GTX_LEAD locs[n] = Loc{2, v, vo, pos - vo};
          ++n;
        }
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// graph walks: Graph::get_labels_forward / get_labels_backward (graph.cpp:1187-1439 / 1441-1701)
// Sequences are never materialised: a candidate keeps its length and the mismatches of its already compared prefix
// (count_mismatches restarts from 0 every time in the reference, which is the same sum).
// ---------------------------------------------------------------------------------------------------------------
struct SubRead
{
  uint8_t const * rd; // codes of the whole read (LDS)
  uint32_t begin;     // first base of the sub-read
  uint32_t len;       // L
};

// count_mismatches / count_mismatches_backward (graph_utils.hpp:7-69) of graph codes dna[0..n) against the sub-read,
// 64 characters per step, one per lane.  forward: dna[i] <-> sub-read[at+i]; backward: the candidate already covers
// the last `at` characters and dna is prepended, dna[n-1-i] <-> sub-read[L-1-at-i].  Comparison stops at the
// sub-read's end.  Returns the running count capped at max+1 (= dead); a '<' / '>' in the compared range kills.
// (The reference stops counting at max+1 or at the tag, whichever comes first -- both mean "rejected".)
template <class W, bool BACKWARD>
GTX_DEV uint32_t cmp_codes(SubRead const & sr, uint32_t at, uint8_t const * dna, uint32_t n, uint32_t mism, uint32_t maxmm)
{
  mism = GTX_U(mism);
  n = GTX_U(n);
  at = GTX_U(at);
  if (mism > maxmm)
    return maxmm + 1;
  uint32_t const room = at < sr.len ? sr.len - at : 0;
  uint32_t const m = n < room ? n : room;
  for (uint32_t base = 0; base < m; base += 64)
  {
    typename W::template PerLane<bool> kill, mm;
    W::lanes([&](uint32_t l) {
      uint32_t const i = base + l;
      bool k = false, x = false;
      if (i < m)
      {
        uint8_t const gc = BACKWARD ? dna[n - 1 - i] : dna[i];
        uint8_t const rc = BACKWARD ? sr.rd[sr.begin + sr.len - 1 - at - i] : sr.rd[sr.begin + at + i];
        k = gc == DNA_KILL;
        x = gc != rc && rc != 15 && gc != 15;
      }
      kill[l] = k;
      mm[l] = x;
    });
    if (W::ballot(kill) != 0)
      return maxmm + 1;
    mism += static_cast<uint32_t>(__builtin_popcountll(W::ballot(mm)));
    if (mism > maxmm)
      return maxmm + 1;
  }
  return mism;
}

// cmp_codes for a piece of at most 64 characters that is already in registers, character k of the compare order in lane k
// (forward: dna[k]; backward: dna[n-1-k]).  Same counting, same kill rule, no memory access besides the read's codes in LDS.
template <class W, bool BACKWARD, class Codes>
GTX_DEV uint32_t cmp_lane_codes(SubRead const & sr, uint32_t at, Codes const & code, uint32_t n, uint32_t mism, uint32_t maxmm)
{
  mism = GTX_U(mism);
  n = GTX_U(n);
  at = GTX_U(at);
  if (mism > maxmm)
    return maxmm + 1;
  uint32_t const room = at < sr.len ? sr.len - at : 0;
  uint32_t const m = n < room ? n : room; // (<= 64: the caller's condition)
  typename W::template PerLane<bool> kill, mm;
  W::lanes([&](uint32_t l) {
    bool k = false, x = false;
    if (l < m)
    {
      uint8_t const gc = static_cast<uint8_t>(code[l]);
      uint8_t const rc = BACKWARD ? sr.rd[sr.begin + sr.len - 1 - at - l] : sr.rd[sr.begin + at + l];
      k = gc == DNA_KILL;
      x = gc != rc && rc != 15 && gc != 15;
    }
    kill[l] = k;
    mm[l] = x;
  });
  if (W::ballot(kill) != 0)
    return maxmm + 1;
  mism += static_cast<uint32_t>(__builtin_popcountll(W::ballot(mm)));
  return mism > maxmm ? maxmm + 1 : mism;
}

// copy of a walk candidate: only the part in use when the entry is long (the exact pass has room for a variant node per base)
template <class W>
GTX_DEV void copy_cand(Cand & dst, Cand const & src)
{
  if constexpr (CAND_WORDS <= 64)
    copy_entry<W>(dst, src);
  else
  {
    if (&dst == &src)
      return;
    uint32_t const words = 4 + GTX_U(src.nids);
    uint32_t * d = reinterpret_cast<uint32_t *>(&dst);
    uint32_t const * s = reinterpret_cast<uint32_t const *>(&src);
    W::lanes([&](uint32_t l) {
      for (uint32_t w = l; w < words; w += 64)
        d[w] = s[w];
    });
    W::lds_sync();
  }
}

template <class W>
GTX_DEV void cand_erase(Cand * c, uint32_t n, uint32_t j)
{
  for (uint32_t k = j; k + 1 < n; ++k)
    copy_cand<W>(c[k], c[k + 1]);
}

// Emits the labels of the candidates that tie the fewest mismatches (graph.cpp:1375-1437 / 1636-1698).
// `fixed_pos` = start position (forward) or end position (backward) shared by all labels of this location.
template <class W, bool BACKWARD>
GTX_DEV bool emit_best(GraphView const & g, Cand const * cand, uint32_t n, uint32_t L, uint32_t fixed_pos, uint32_t & max_mismatches,
                       DevLabel * out, uint32_t & n_out, uint32_t out_cap, uint32_t & status)
{
  uint32_t const first_out = n_out;
  constexpr uint32_t PENDING = 0xFFFFFFFFu; // (no allele number reaches this: a site has at most a few thousand)
  for (uint32_t j = 0; j < n; ++j)
  {
    if (GTX_U(cand[j].len) < L)
      continue;
    uint32_t const mm = GTX_U(cand[j].mism);
    if (mm > max_mismatches)
      continue;
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      n_out = first_out;
    }
    uint32_t const nids = GTX_U(cand[j].nids);
    uint32_t const nl = nids == 0 ? 1 : nids;
    if (n_out + nl > out_cap)
    {
      status |= GTX_ST_DFS_OVERFLOW;
      return false;
    }
    uint32_t const p = GTX_U(cand[j].pos);
    uint32_t const s = BACKWARD ? p : fixed_pos, e = BACKWARD ? fixed_pos : p;
    if (nids == 0)
    {
      GTX_LEAD out[n_out] = DevLabel{s, e, INVALID, 0};
      ++n_out;
    }
    else
    {
      // the variant node of every label for now (allele = PENDING); site and allele number are looked up for all labels
      // together below -- two dependent fetches in all instead of two per label, one after the other
      W::lanes([&](uint32_t l) {
        for (uint32_t k = l; k < nids; k += 64)
          out[n_out + k] = DevLabel{s, e, cand[j].ids[k], PENDING};
      });
      n_out += nids;
    }
  }
  W::lds_sync();
  for (uint32_t b = first_out; b < n_out; b += 64)
    W::lanes([&](uint32_t l) {
      uint32_t const k = b + l;
      if (k < n_out && out[k].allele == PENDING && out[k].site != INVALID)
      {
        uint32_t const v = out[k].site;
        uint32_t const vs = g.var_out_ref[v] - 1;
        out[k].site = vs;
        out[k].allele = v - g.ref_first_var[vs];
      }
    });
  W::lds_sync();
  return true;
}

// One start (forward) or end (backward) location: extends over variant sites until every candidate covers the
// sub-read, keeps candidates within the mismatch budget, appends the labels of the best ones to `out`.
template <class W, bool BACKWARD>
GTX_DEV bool labels_walk(GraphView const & g, Loc const & s, SubRead const & sr, uint32_t & max_mismatches, Cand * cand, uint32_t cand_cap,
                         DevLabel * out, uint32_t & n_out, uint32_t out_cap, uint32_t & status)
{
  uint8_t const * dna = reinterpret_cast<uint8_t const *>(g.dna);
  uint32_t const L = sr.len;
  uint32_t const maxmm = max_mismatches;
  uint32_t n = 1;
  uint32_t site = INVALID; // site whose alleles come next, INVALID = none
  uint32_t const s_type = GTX_U(s.type), s_node = GTX_U(s.node), s_offset = GTX_U(s.offset), s_order = GTX_U(s.order);
  {
    uint32_t len, mism, pos, nids = 0, id0 = 0;
    if (s_type == 2)
    {
      uint32_t const v = s_node;
      nids = 1;
      id0 = v;
      uint32_t const vout = GTX_U(g.var_out_ref[v]);
      uint32_t const vsite = vout - 1;
      uint32_t const vo = GTX_U(g.var_order[v]), vl = GTX_U(g.var_len[v]), vd = GTX_U(g.var_dna[v]);
      // (the tables of the reference node that follows / precedes the allele are fetched before the allele is compared: one
      // round trip instead of one per table behind the compare)
      uint32_t const r = BACKWARD ? vsite : vout;
      uint32_t const rl = GTX_U(g.ref_len[r]), rdo = GTX_U(g.ref_dna[r]), rord = GTX_U(g.ref_order[r]), rnv = GTX_U(g.ref_nvar[r]);
      if (!BACKWARD)
      {
        len = vl - s_offset;
        mism = cmp_codes<W, false>(sr, 0, dna + vd + s_offset, len, 0, maxmm);
        if (len >= L)
          pos = ug_special_of<W>(g, vsite, (vo + vl - 1) - (len - L));
        else
        {
          mism = cmp_codes<W, false>(sr, len, dna + rdo, rl, mism, maxmm);
          len += rl;
          pos = (rord + rl - 1) - (len - L);
          if (rnv > 0)
            site = r;
        }
      }
      else
      {
        len = s_offset + 1;
        mism = cmp_codes<W, true>(sr, 0, dna + vd, len, 0, maxmm);
        if (len >= L)
          pos = ug_special_of<W>(g, vsite, vo + (len - L));
        else
        {
          mism = cmp_codes<W, true>(sr, len, dna + rdo, rl, mism, maxmm);
          len += rl;
          pos = rord + (len - L);
          if (r != 0)
            site = r - 1;
        }
      }
    }
    else
    {
      uint32_t const r = s_node;
      uint32_t const rl = GTX_U(g.ref_len[r]), rd_ = GTX_U(g.ref_dna[r]), rnv = GTX_U(g.ref_nvar[r]); // (one round trip)
      if (!BACKWARD)
      {
        len = rl - s_offset;
        mism = cmp_codes<W, false>(sr, 0, dna + rd_ + s_offset, len, 0, maxmm);
        pos = (s_order + rl - 1) - (len - L); // s_order is the node's order
        if (rnv > 0)
          site = r;
      }
      else
      {
        if (r != 0)
          site = r - 1;
        len = s_offset + 1;
        mism = cmp_codes<W, true>(sr, 0, dna + rd_, len, 0, maxmm);
        pos = s_order + (len - L);
      }
    }
    GTX_LEAD
    {
      cand[0].len = len;
      cand[0].mism = mism;
      cand[0].pos = pos;
      cand[0].nids = nids;
      cand[0].ids[0] = id0;
    }
    W::lds_sync();
  }

  if (site != INVALID && GTX_U(cand[0].len) < L)
  {
    bool all_long = false;
    while (!all_long && n < 128 && site != INVALID)
    {
      all_long = true;
      uint32_t const r = BACKWARD ? site : site + 1; // reference node appended (forward) / prepended (backward)
      uint32_t const fv = GTX_U(g.ref_first_var[site]), nv = GTX_U(g.ref_nvar[site]);
      uint8_t const * rdna = dna + GTX_U(g.ref_dna[r]);
      uint32_t const rlen = GTX_U(g.ref_len[r]);
      uint32_t const rorder = GTX_U(g.ref_order[r]);
      // (with them, in the same round trip: what the end of the round and the special positions of the site ask for)
      uint32_t const r_nvar = GTX_U(g.ref_nvar[r]);
      uint32_t const s_reach = GTX_U(g.site_ref_reach[site]), s_base = GTX_U(g.site_special_base[site]);
      auto special_of = [&](uint32_t p) { return p > s_reach ? SPECIAL_START + s_base + (p - s_reach - 1) : p; }; // (g_special_of)
      uint32_t original = n;
      // The site's tables and bases once, in two round trips for all alleles together (lane i: allele i's order, length and
      // up to WALK_ALLELE_BYTES of its bases in compare order; lane k: character k of the reference node behind / in front
      // of the site), instead of three dependent fetches per (candidate, allele) inside the loops below -- on a graph of
      // merged multi-allelic sites the walk at the read's end was half of the general pass.
      constexpr uint32_t WALK_ALLELE_BYTES = 16;
      bool const staged = nv <= 64;
      typename W::template PerLane<uint32_t> a_order, a_len, a_dna, a_b0, a_b1, a_b2, a_b3, r_code;
      W::lanes([&](uint32_t l) {
        uint32_t vo = 0, vl = 0, vd = 0;
        if (staged && l < nv)
        {
          vo = g.var_order[fv + l];
          vl = g.var_len[fv + l];
          vd = g.var_dna[fv + l];
        }
        a_order[l] = vo;
        a_len[l] = vl;
        a_dna[l] = vd;
        r_code[l] = l < rlen ? static_cast<uint32_t>(BACKWARD ? rdna[rlen - 1 - l] : rdna[l]) : 0u;
      });
      W::lanes([&](uint32_t l) {
        uint32_t b[4] = {0, 0, 0, 0};
        uint32_t const vl = a_len[l], vd = a_dna[l];
        if (staged && l < nv && vl <= WALK_ALLELE_BYTES)
          for (uint32_t k = 0; k < WALK_ALLELE_BYTES; ++k)
            if (k < vl)
              b[k >> 2] |= static_cast<uint32_t>(BACKWARD ? dna[vd + vl - 1 - k] : dna[vd + k]) << (8 * (k & 3u));
        a_b0[l] = b[0];
        a_b1[l] = b[1];
        a_b2[l] = b[2];
        a_b3[l] = b[3];
      });
      for (uint32_t j = 0; j < original; ++j)
      {
        uint32_t const jlen = GTX_U(cand[j].len), jmism = GTX_U(cand[j].mism), jn = GTX_U(cand[j].nids);
        if (jlen >= L)
          continue;
        for (uint32_t i = 0; i < nv; ++i)
        {
          bool const last = i + 1 == nv; // the last allele extends candidate j in place, the others branch off copies
          uint32_t const v = fv + i;
          uint32_t const vo = staged ? W::from_lane(a_order, i) : GTX_U(g.var_order[v]);
          uint32_t const vl = staged ? W::from_lane(a_len, i) : GTX_U(g.var_len[v]);
          uint32_t len = jlen;
          uint32_t mm;
          if (staged && vl <= WALK_ALLELE_BYTES)
          {
            uint32_t const w0 = W::from_lane(a_b0, i), w1 = W::from_lane(a_b1, i), w2 = W::from_lane(a_b2, i), w3 = W::from_lane(a_b3, i);
            typename W::template PerLane<uint32_t> a_code;
            W::lanes([&](uint32_t l) {
              uint32_t const w = l < 4 ? w0 : l < 8 ? w1 : l < 12 ? w2 : w3;
              a_code[l] = (w >> (8 * (l & 3u))) & 255u;
            });
            mm = cmp_lane_codes<W, BACKWARD>(sr, len, a_code, vl, jmism, maxmm);
          }
          else
            mm = cmp_codes<W, BACKWARD>(sr, len, dna + (staged ? W::from_lane(a_dna, i) : GTX_U(g.var_dna[v])), vl, jmism, maxmm);
          len += vl;
          bool const enough = len >= L;
          if (!enough)
          {
            // (what is compared of the reference node: its first min(rlen, L - len) characters)
            if ((rlen < L - len ? rlen : L - len) <= 64)
              mm = cmp_lane_codes<W, BACKWARD>(sr, len, r_code, rlen, mm, maxmm);
            else
              mm = cmp_codes<W, BACKWARD>(sr, len, rdna, rlen, mm, maxmm);
            len += rlen;
          }
          if (mm <= maxmm)
          {
            if ((!last && n >= cand_cap) || jn >= AlignCfg::MAXIDS)
            {
              status |= GTX_ST_DFS_OVERFLOW;
              return false;
            }
            uint32_t pos;
            if (!BACKWARD)
              pos = enough ? special_of((vo + vl - 1) - (len - L)) : (rorder + rlen - 1) - (len - L);
            else
              pos = enough ? special_of(vo + (len - L)) : rorder + (len - L);
            uint32_t const dst = last ? j : n;
            if (!last)
            {
              copy_cand<W>(cand[n], cand[j]);
              ++n;
            }
            GTX_LEAD
            {
              Cand & nc = cand[dst];
              nc.ids[jn] = v;
              nc.nids = jn + 1;
              nc.len = len;
              nc.mism = mm;
              nc.pos = pos;
            }
            W::lds_sync();
            if (len < L)
              all_long = false;
          }
          else if (last)
          {
            cand_erase<W>(cand, n, j);
            --n;
            --original;
            --j;
          }
        }
      }
      if (all_long)
        break;
      if (!BACKWARD)
        site = r_nvar > 0 ? r : INVALID;
      else
      {
        if (r == 0)
          break;
        site = r - 1;
      }
    }
  }

  uint32_t fixed = s_order + s_offset;
  if (s_type == 2)
    fixed = ug_special_of<W>(g, GTX_U(g.var_out_ref[s_node]) - 1, fixed);
  return emit_best<W, BACKWARD>(g, cand, n, L, fixed, max_mismatches, out, n_out, out_cap, status);
}

// Graph::iterative_dfs (graph.cpp:1703-1754): labels of all locations that tie the fewest mismatches
template <class W>
GTX_DEV uint32_t iterative_dfs(GraphView const & g, AlignWorkspace & ws, uint32_t n_locs, bool backward, SubRead const & sr,
                               uint32_t & max_mismatches, uint32_t & status)
{
  uint32_t n_out = 0;
  if (n_locs > 1024)
    return 0;
  WalkBuffers & wb = ws.u.w;
  for (uint32_t k = 0; k < n_locs; ++k)
  {
    uint32_t mm = max_mismatches;
    uint32_t const before = n_out;
    uint32_t after = n_out;
    bool const ok = backward ? labels_walk<W, true>(g, wb.locs[k], sr, mm, wb.cand, cap_cand(ws), wb.dfs_out, after, cap_wl(ws), status)
                             : labels_walk<W, false>(g, wb.locs[k], sr, mm, wb.cand, cap_cand(ws), wb.dfs_out, after, cap_wl(ws), status);
    if (!ok)
      return 0;
    if (after == before)
      continue; // no labels from this location
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      uint32_t const cnt = after - before; // labels = new_labels
      if (before != 0)
        for (uint32_t i = 0; i < cnt; ++i)
          copy_entry<W>(wb.dfs_out[i], wb.dfs_out[before + i]);
      n_out = cnt;
    }
    else if (mm == max_mismatches)
      n_out = after;
    // mm > max_mismatches cannot happen: a walk never returns labels above its budget
  }
  return n_out;
}

// ---------------------------------------------------------------------------------------------------------------
// seed chaining: GenotypePaths::add_next_kmer_labels / add_prev_kmer_labels (genotype_paths.cpp:294-352 / 233-292)
// ---------------------------------------------------------------------------------------------------------------

// find_all_nonduplicated_paths (genotype_paths.cpp:32-66) + Path::merge_with_current (path.cpp:105-129)
// first entry j of a dense table with key[j] == want (exact pass: 64 entries per step, one per lane); n when there is none
template <class W>
GTX_DEV uint32_t find_pair(uint32_t const * a, uint32_t const * b, uint32_t n, uint32_t want_a, uint32_t want_b)
{
  for (uint32_t base = 0; base < n; base += 64)
  {
    typename W::template PerLane<bool> hit;
    W::lanes([&](uint32_t l) { hit[l] = base + l < n && a[base + l] == want_a && b[base + l] == want_b; });
    uint64_t const m = W::ballot(hit);
    if (m != 0)
      return base + static_cast<uint32_t>(__builtin_ctzll(m));
  }
  return n;
}

template <class W, class WS, class PP>
GTX_DEV uint32_t make_pp(Here, WS & ws, PP & pp, DevLabel const * ll, uint32_t n, uint32_t rs, uint32_t re, uint32_t mism, uint32_t & status)
{
  uint32_t npp = 0;
  for (uint32_t i = 0; i < n; ++i)
  {
    uint32_t const ls = GTX_U(ll[i].start), le = GTX_U(ll[i].end), lsite = GTX_U(ll[i].site), lall = GTX_U(ll[i].allele);
    if (lsite != INVALID && !pv_fits(lall))
    {
      status |= GTX_ST_PATH_OVERFLOW | GTX_ST_WIDE_ALLELE;
      return npp;
    }
    uint32_t d = 0;
    if constexpr (DENSE_PP_KEYS)
      d = find_pair<W>(ws.pp_start, ws.pp_end, npp, ls, le);
    else
      for (; d < npp; ++d)
        if (GTX_U(pp[d].start) == ls && GTX_U(pp[d].end) == le)
          break;
    if (d == npp)
    {
      if (npp >= cap_pp(ws))
      {
        status |= GTX_ST_PATH_OVERFLOW;
        return npp;
      }
      GTX_LEAD
      {
        if constexpr (DENSE_PP_KEYS)
        {
          ws.pp_start[npp] = ls;
          ws.pp_end[npp] = le;
        }
        DPath & p = pp[npp];
        p.start = ls;
        p.end = le;
        p.rs = static_cast<uint16_t>(rs);
        p.re = static_cast<uint16_t>(re);
        p.mism = static_cast<uint16_t>(mism);
        p.nvar = lsite != INVALID ? 1 : 0;
        if (lsite != INVALID)
        {
          p.v[0].site = lsite;
          pv_set_single(p.v[0], lall);
        }
      }
      W::lds_sync();
      ++npp;
      continue;
    }
    if (lsite == INVALID)
      continue;
    DPath & p = pp[d];
    uint32_t const nvar = GTX_U(static_cast<uint32_t>(p.nvar));
    uint32_t k = 0;
    for (; k < nvar; ++k)
      if (GTX_U(p.v[k].site) == lsite)
        break;
    if (k == nvar && nvar >= cap_sites(ws))
    {
      status |= GTX_ST_PATH_OVERFLOW;
      return npp;
    }
    GTX_LEAD
    {
      if (k < nvar)
        pv_add(p.v[k], lall);
      else
      {
        p.v[nvar].site = lsite;
        pv_set_single(p.v[nvar], lall);
        p.nvar = static_cast<uint16_t>(nvar + 1);
      }
    }
    W::lds_sync();
  }
  return npp;
}

// Path::Path(p1, p2) (path.cpp:38-82) into `np` (LDS): everything from p2, allele sets of shared sites intersected
// with p1's, p1's other sites appended, start/read_start_index taken from p1.  false <=> the reference returns early
// on an empty intersection (its caller then discards the half merged object).
template <class W>
GTX_DEV bool merge_paths(DPath const & p1, DPath const & p2, DPath & np, uint32_t max_sites, uint32_t & status)
{
  copy_path<W>(np, p2);
  uint32_t const n1 = GTX_U(static_cast<uint32_t>(p1.nvar));
  uint32_t nn = GTX_U(static_cast<uint32_t>(np.nvar));
  for (uint32_t i = 0; i < n1; ++i)
  {
    uint32_t const s1 = GTX_U(p1.v[i].site);
    uint32_t j = 0;
    for (; j < nn; ++j)
      if (GTX_U(np.v[j].site) == s1)
        break;
    if (j < nn)
    {
      if (!pv_meet_any<W>(np.v[j], p1.v[i]))
        return false;
      GTX_LEAD pv_meet(np.v[j], p1.v[i]);
    }
    else
    {
      if (nn >= max_sites)
      {
        status |= GTX_ST_PATH_OVERFLOW;
        return false;
      }
      GTX_LEAD np.v[nn] = p1.v[i];
      ++nn;
    }
    W::lds_sync();
  }
  GTX_LEAD
  {
    np.nvar = static_cast<uint16_t>(nn);
    np.rs = p1.rs;
    np.start = p1.start;
    np.mism = static_cast<uint16_t>(np.mism + p1.mism);
  }
  W::lds_sync();
  return true;
}

// appends to ws.paths; n_paths is tracked by the caller
template <class W>
GTX_DEV bool push_path(AlignWorkspace & ws, uint32_t & n_paths, DPath const & p, uint32_t & status)
{
  if (n_paths >= cap_paths(ws))
  {
    status |= GTX_ST_PATH_OVERFLOW;
    return false;
  }
  copy_path<W>(ws.paths[n_paths], p);
  ++n_paths;
  return true;
}

// add_next_kmer_labels for the list that brings the passes with tables in HBM their work: the hundreds of places of one exact
// k-mer inside a long repeat.  When no label carries a variant site and the labels come in ascending start order (the order
// of the index' sweep for one key) the general code below does nothing but this: a path that ends where a label starts is
// extended by it in place -- Path(p1, p2) of a site-free p2 is p1 with p2's end, read end and mismatches --, at most one label
// per path (the starts are distinct), and the labels no path took are appended as new paths in label order.  Every step of
// that is independent per path / per label, so it is done 64 at a time: a binary search of the path's end in the sorted
// starts per lane, one store of the changed fields, an atomic OR into the set of taken labels, a prefix sum for the places of
// the appended paths.  Returns false -- nothing touched -- when the list is not of that kind.
template <class W, class WS>
GTX_DEV bool chain_in_bulk(Here, WS & ws, DevLabel const * ll, uint32_t n, uint32_t rs, uint32_t re, uint32_t mism, uint32_t & n_paths,
                           uint32_t & longest, uint32_t & status)
{
  if constexpr (!DENSE_PP_KEYS)
    return false;
  else
  {
    for (uint32_t base = 0; base < n; base += 64)
    {
      typename W::template PerLane<bool> bad;
      W::lanes([&](uint32_t l) {
        uint32_t const k = base + l;
        bad[l] = k < n && (ll[k].site != INVALID || (k + 1 < n && !(ll[k].start < ll[k + 1].start)));
      });
      if (W::ballot(bad) != 0)
        return false;
    }
    if (n > cap_pp(ws))
    {
      status |= GTX_ST_PATH_OVERFLOW;
      return true;
    }
    uint32_t * const starts = ws.pp_start, * const ends = ws.pp_end;
    for (uint32_t base = 0; base < n; base += 64)
      W::lanes([&](uint32_t l) {
        if (base + l < n)
        {
          starts[base + l] = ll[base + l].start;
          ends[base + l] = ll[base + l].end;
        }
      });
    MemBits taken = bits_clear<W>(ws.bits_pp, n); // (ends with the sync the tables above need as well)
    uint32_t const original_size = n_paths;
    for (uint32_t base = 0; base < original_size; base += 64)
    {
      typename W::template PerLane<uint32_t> size;
      W::lanes([&](uint32_t l) {
        uint32_t const i = base + l;
        uint32_t sz = 0;
        if (i < original_size)
        {
          DPath & p = ws.paths[i];
          if (static_cast<uint32_t>(p.re) == rs)
          {
            uint32_t const want = p.end;
            uint32_t lo = 0, hi = n;
            while (lo < hi)
            {
              uint32_t const mid = (lo + hi) >> 1;
              if (starts[mid] < want)
                lo = mid + 1;
              else
                hi = mid;
            }
            if (lo < n && starts[lo] == want)
            {
              p.end = ends[lo];
              p.re = static_cast<uint16_t>(re);
              p.mism = static_cast<uint16_t>(p.mism + mism);
              W::atomic_or_u64(taken.w + (lo >> 6), 1ull << (lo & 63u));
              sz = re - static_cast<uint32_t>(p.rs) + 1u;
            }
          }
        }
        size[l] = sz;
      });
      uint32_t const m = W::max(size);
      if (m > longest)
        longest = m;
    }
    W::lds_sync();
    uint32_t n_left = 0;
    for (uint32_t base = 0; base < n; base += 64)
    {
      uint64_t const valid = n - base >= 64 ? ~0ull : ((1ull << (n - base)) - 1ull);
      n_left += static_cast<uint32_t>(__builtin_popcountll(~GTX_U(taken.w[base >> 6]) & valid));
    }
    if (n_left == 0)
      return true;
    if (n_paths + n_left > cap_paths(ws))
    {
      status |= GTX_ST_PATH_OVERFLOW;
      return true;
    }
    for (uint32_t base = 0; base < n; base += 64)
    {
      uint64_t const valid = n - base >= 64 ? ~0ull : ((1ull << (n - base)) - 1ull);
      uint64_t const left = ~GTX_U(taken.w[base >> 6]) & valid;
      if (left == 0)
        continue;
      W::lanes([&](uint32_t l) {
        if ((left >> l) & 1ull)
        {
          DPath & p = ws.paths[n_paths + static_cast<uint32_t>(__builtin_popcountll(left & ((1ull << l) - 1ull)))];
          p.start = starts[base + l];
          p.end = ends[base + l];
          p.rs = static_cast<uint16_t>(rs);
          p.re = static_cast<uint16_t>(re);
          p.mism = static_cast<uint16_t>(mism);
          p.nvar = 0;
        }
      });
      n_paths += static_cast<uint32_t>(__builtin_popcountll(left));
    }
    W::lds_sync();
    if (re - rs + 1u > longest)
      longest = re - rs + 1u;
    return true;
  }
}

template <class W>
GTX_DEV void add_kmer_labels(AlignWorkspace & ws, DevLabel const * ll, uint32_t n, uint32_t rs, uint32_t re, uint32_t mism,
                             bool prev, uint32_t & n_paths, uint32_t & longest, uint32_t & status)
{
  if (n == 0)
    return;
  if (n >= 16 && !prev && chain_in_bulk<W>(Here{}, ws, ll, n, rs, re, mism, n_paths, longest, status))
    return;
  if (n == 1 && !prev)
  {
    // One label = one new path P with at most one variant site.  Specialisation of the general code below: P can only
    // ever replace in place (a path merges with the single P at most once), and Path(p1, P) is p1 with its end, read
    // end and mismatches advanced and P's site moved to the front of the site list (path.cpp:38-82 keeps p2's sites
    // first), its allele set intersected when p1 already carries the site.
    uint32_t const ls = GTX_U(ll[0].start), le = GTX_U(ll[0].end), lsite = GTX_U(ll[0].site), lall = GTX_U(ll[0].allele);
    if (lsite != INVALID && !pv_fits(lall))
    {
      status |= GTX_ST_PATH_OVERFLOW | GTX_ST_WIDE_ALLELE;
      return;
    }
    bool matched = false;
    uint32_t const original_size = n_paths;
    // one path that ends where the label starts (false: a table is full)
    auto extend = [&](uint32_t i) -> bool
    {
      DPath & p = ws.paths[i];
      uint32_t const w2 = GTX_U(reinterpret_cast<uint32_t const *>(&p)[2]); // rs | re << 16
      if ((w2 >> 16) != rs || GTX_U(p.end) != ls)
        return true;
      uint32_t const nvar = GTX_U(static_cast<uint32_t>(p.nvar));
      uint32_t j = nvar;
      if (lsite != INVALID)
      {
        for (j = 0; j < nvar; ++j)
          if (GTX_U(p.v[j].site) == lsite)
            break;
        if (j < nvar)
        {
          if (!pv_has<W>(p.v[j], lall))
            return true; // empty allele intersection: this path does not merge
        }
        else if (nvar >= cap_sites(ws))
        {
          status |= GTX_ST_PATH_OVERFLOW;
          return false;
        }
      }
      GTX_LEAD
      {
        if (lsite != INVALID)
        {
          for (uint32_t k = j; k > 0; --k) // sites before j (or all of them) move one place back
            p.v[k] = p.v[k - 1];
          p.v[0].site = lsite;
          pv_set_single(p.v[0], lall); // ({allele} met with a set that holds it)
          if (j == nvar)
            p.nvar = static_cast<uint16_t>(nvar + 1);
        }
        p.end = le;
        p.re = static_cast<uint16_t>(re);
        p.mism = static_cast<uint16_t>(p.mism + mism);
      }
      W::lds_sync();
      matched = true;
      uint32_t const sz = re - (w2 & 0xFFFFu) + 1u;
      if (sz > longest)
        longest = sz;
      return true;
    };
    if constexpr (DENSE_PP_KEYS)
    {
      // (hundreds of paths in HBM, and this is called once per label list of a walk: the paths that end at the label's start
      //  are found 64 at a time -- one strided load per lane -- instead of one round trip to HBM per path)
      for (uint32_t base = 0; base < original_size; base += 64)
      {
        typename W::template PerLane<bool> hit;
        W::lanes([&](uint32_t l) {
          uint32_t const i = base + l;
          bool h = false;
          if (i < original_size)
          {
            DPath const & p = ws.paths[i];
            h = static_cast<uint32_t>(p.re) == rs && p.end == ls;
          }
          hit[l] = h;
        });
        for (uint64_t m = W::ballot(hit); m != 0; m &= m - 1)
          if (!extend(base + static_cast<uint32_t>(__builtin_ctzll(m))))
            return;
      }
    }
    else
      for (uint32_t i = 0; i < original_size; ++i)
        if (!extend(i))
          return;
    if (!matched)
    {
      if (n_paths >= cap_paths(ws))
      {
        status |= GTX_ST_PATH_OVERFLOW;
        return;
      }
      GTX_LEAD
      {
        DPath & p = ws.paths[n_paths];
        p.start = ls;
        p.end = le;
        p.rs = static_cast<uint16_t>(rs);
        p.re = static_cast<uint16_t>(re);
        p.mism = static_cast<uint16_t>(mism);
        p.nvar = lsite != INVALID ? 1 : 0;
        p.v[0].site = lsite;
        if (lsite != INVALID)
          pv_set_single(p.v[0], lall);
        else
          pv_clear(p.v[0]);
      }
      W::lds_sync();
      ++n_paths;
      uint32_t const sz = re - rs + 1u;
      if (sz > longest)
        longest = sz;
    }
    return;
  }
  auto & pp = ws.u.w.pp;
  uint32_t const npp = make_pp<W>(Here{}, ws, pp, ll, n, rs, re, mism, status);
  if (status)
    return;
  uint32_t const original_size = n_paths;
  auto matched = new_pp_set<W>(ws, npp);
  // one path whose read end (start) is where the new labels begin (end); false: a table is full
  auto chain = [&](uint32_t i) -> bool
  {
    if (prev ? (GTX_U(static_cast<uint32_t>(ws.paths[i].rs)) != re) : (GTX_U(static_cast<uint32_t>(ws.paths[i].re)) != rs))
      return true;
    bool once = false;
    uint32_t const o_start = GTX_U(ws.paths[i].start), o_end = GTX_U(ws.paths[i].end);
    if constexpr (DENSE_PP_KEYS)
    {
      // how many pp entries abut this path, and the first of them
      uint32_t const * key = pp_keys(ws, prev);
      uint32_t const want = prev ? o_start : o_end;
      uint32_t n_hit = 0, first = 0;
      for (uint32_t base = 0; base < npp; base += 64)
      {
        typename W::template PerLane<bool> hit;
        W::lanes([&](uint32_t l) { hit[l] = base + l < npp && key[base + l] == want; });
        uint64_t const m = W::ballot(hit);
        if (m != 0 && n_hit == 0)
          first = base + static_cast<uint32_t>(__builtin_ctzll(m));
        n_hit += static_cast<uint32_t>(__builtin_popcountll(m));
      }
      if (n_hit == 0)
        return true; // (nothing to copy, nothing to merge)
      if (n_hit == 1 && !prev && GTX_U(static_cast<uint32_t>(ws.paths[i].nvar)) == 0 && GTX_U(static_cast<uint32_t>(pp[first].nvar)) == 0)
      {
        // The one abutting entry, and neither side carries a variant site (chains inside a reference node: the repeats
        // that bring hundreds of paths here): Path(p1, p2) is p1 with p2's end, read end and mismatches -- in place, no copies
        uint32_t const p_end = GTX_U(pp[first].end), p_re = GTX_U(static_cast<uint32_t>(pp[first].re)), p_mm = GTX_U(static_cast<uint32_t>(pp[first].mism));
        uint32_t const o_rs = GTX_U(static_cast<uint32_t>(ws.paths[i].rs));
        GTX_LEAD
        {
          DPath & p = ws.paths[i];
          p.end = p_end;
          p.re = static_cast<uint16_t>(p_re);
          p.mism = static_cast<uint16_t>(p.mism + p_mm);
        }
        bits_set<W>(matched, first);
        W::lds_sync();
        uint32_t const sz = p_re - o_rs + 1u;
        if (sz > longest)
          longest = sz;
        return true;
      }
    }
    copy_path<W>(ws.orig, ws.paths[i]);
    // one pp entry that abuts the path: merged into it (the first one in place, further ones as new paths)
    auto join = [&](uint32_t j) -> bool
    {
      bool const ok = prev ? merge_paths<W>(pp[j], ws.orig, ws.np, cap_sites(ws), status) : merge_paths<W>(ws.orig, pp[j], ws.np, cap_sites(ws), status);
      if (status)
        return false;
      if (!ok)
        return true;
      bits_set<W>(matched, j);
      if (once)
        return push_path<W>(ws, n_paths, ws.np, status);
      uint32_t const sz = upath_size<W>(ws.np);
      if (sz > longest)
        longest = sz;
      copy_path<W>(ws.paths[i], ws.np);
      once = true;
      return true;
    };
    if constexpr (DENSE_PP_KEYS)
    {
      // (hundreds to thousands of entries: the abutting ones are found 64 at a time in the dense start / end table, then taken in order)
      uint32_t const * key = pp_keys(ws, prev);
      uint32_t const want = prev ? o_start : o_end;
      for (uint32_t base = 0; base < npp; base += 64)
      {
        typename W::template PerLane<bool> hit;
        W::lanes([&](uint32_t l) { hit[l] = base + l < npp && key[base + l] == want; });
        for (uint64_t m = W::ballot(hit); m != 0; m &= m - 1)
          if (!join(base + static_cast<uint32_t>(__builtin_ctzll(m))))
            return false;
      }
    }
    else
      for (uint32_t j = 0; j < npp; ++j)
      {
        if (prev ? !(GTX_U(pp[j].end) == o_start) : !(o_end == GTX_U(pp[j].start)))
          continue;
        if (!join(j))
          return false;
      }
    return true;
  };
  if constexpr (DENSE_PP_KEYS)
  {
    for (uint32_t base = 0; base < original_size; base += 64) // (the paths at the right read index, 64 at a time)
    {
      typename W::template PerLane<bool> hit;
      W::lanes([&](uint32_t l) {
        uint32_t const i = base + l;
        hit[l] = i < original_size && (prev ? static_cast<uint32_t>(ws.paths[i].rs) == re : static_cast<uint32_t>(ws.paths[i].re) == rs);
      });
      for (uint64_t m = W::ballot(hit); m != 0; m &= m - 1)
        if (!chain(base + static_cast<uint32_t>(__builtin_ctzll(m))))
          return;
    }
  }
  else
    for (uint32_t i = 0; i < original_size; ++i)
      if (!chain(i))
        return;
  for (uint32_t j = 0; j < npp; ++j)
    if (!bits_get<W>(matched, j))
    {
      uint32_t const sz = upath_size<W>(pp[j]);
      if (sz > longest)
        longest = sz;
      if (!push_path<W>(ws, n_paths, pp[j], status))
        return;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// path filters (genotype_paths.cpp)
// ---------------------------------------------------------------------------------------------------------------

// The passes whose tables are in HBM (DENSE_PP_KEYS) meet hundreds to thousands of paths, and a loop that visits them one by
// one is a round trip to HBM per path: there the filters look at 64 paths per step, one per lane.
// mark_paths: the set of the paths for which `pred(path)` (evaluated by the path's lane) holds, and their number.
template <class W, class Pred>
GTX_DEV auto mark_paths(AlignWorkspace & ws, uint32_t n_paths, uint32_t & n_marked, Pred && pred)
{
  auto set = new_path_set<W>(ws, n_paths);
  n_marked = 0;
  for (uint32_t base = 0; base < n_paths; base += 64)
  {
    typename W::template PerLane<bool> hit;
    W::lanes([&](uint32_t l) {
      uint32_t const i = base + l;
      hit[l] = i < n_paths && pred(ws.paths[i]);
    });
    uint64_t const m = W::ballot(hit);
    if (m != 0)
    {
      bits_set_word<W>(set, base >> 6, m);
      n_marked += static_cast<uint32_t>(__builtin_popcountll(m));
    }
  }
  return set;
}

// stable removal of the paths whose bit in `drop` is set (n_drop of them)
template <class W, class PathSet>
GTX_DEV uint32_t compact_paths(AlignWorkspace & ws, uint32_t n_paths, PathSet const & drop, uint32_t n_drop)
{
  if (n_drop == 0)
    return n_paths;
  uint32_t k = 0;
  if constexpr (DENSE_PP_KEYS)
  {
    // 64 paths per step: the survivors of a step that carry no variant site (16 bytes each) move together -- every lane holds
    // its path before any lane writes, and a survivor never moves behind its place --, a step with a path that has sites
    // moves its survivors one by one
    for (uint32_t base = 0; base < n_paths; base += 64)
    {
      typename W::template PerLane<bool> keep_l, plain_l;
      typename W::template PerLane<uint32_t> w0, w1, w2, w3;
      uint64_t const dropped = bits_word<W>(drop, base >> 6);
      W::lanes([&](uint32_t l) {
        uint32_t const i = base + l;
        bool keep = false, plain = true;
        uint32_t a = 0, b = 0, c = 0, d = 0;
        if (i < n_paths && !((dropped >> l) & 1ull))
        {
          keep = true;
          uint32_t const * src = reinterpret_cast<uint32_t const *>(&ws.paths[i]);
          a = src[0];
          b = src[1];
          c = src[2];
          d = src[3];
          plain = (d >> 16) == 0; // nvar
        }
        keep_l[l] = keep;
        plain_l[l] = plain;
        w0[l] = a;
        w1[l] = b;
        w2[l] = c;
        w3[l] = d;
      });
      uint64_t const keep = W::ballot(keep_l);
      if (keep == 0)
        continue;
      uint32_t const n_keep = static_cast<uint32_t>(__builtin_popcountll(keep));
      if (k == base && n_keep == (n_paths - base < 64 ? n_paths - base : 64u))
      {
        k += n_keep; // (nothing dropped so far: the paths are where they belong)
        continue;
      }
      if (W::ballot(plain_l) == ~0ull)
      {
        W::lanes([&](uint32_t l) {
          if (keep_l[l])
          {
            uint32_t * dst = reinterpret_cast<uint32_t *>(&ws.paths[k + static_cast<uint32_t>(__builtin_popcountll(keep & ((1ull << l) - 1ull)))]);
            dst[0] = w0[l];
            dst[1] = w1[l];
            dst[2] = w2[l];
            dst[3] = w3[l];
          }
        });
        W::lds_sync();
        k += n_keep;
      }
      else
        for (uint64_t m = keep; m != 0; m &= m - 1)
        {
          uint32_t const i = base + static_cast<uint32_t>(__builtin_ctzll(m));
          if (k != i)
            copy_path<W>(ws.paths[k], ws.paths[i]);
          ++k;
        }
    }
    return k;
  }
  for (uint32_t i = 0; i < n_paths; ++i)
    if (!bits_get<W>(drop, i))
    {
      if (k != i)
        copy_path<W>(ws.paths[k], ws.paths[i]);
      ++k;
    }
  return k;
}

template <class W>
GTX_DEV uint32_t remove_short_paths(AlignWorkspace & ws, uint32_t n_paths, uint32_t longest) // :824-834
{
  if (longest <= 1)
    return n_paths;
  if constexpr (DENSE_PP_KEYS)
  {
    uint32_t n_drop;
    auto const drop = mark_paths<W>(ws, n_paths, n_drop, [&](DPath const & p) { return path_size(p) < longest; });
    return compact_paths<W>(ws, n_paths, drop, n_drop);
  }
  auto drop = new_path_set<W>(ws, n_paths);
  uint32_t n_drop = 0;
  for (uint32_t i = 0; i < n_paths; ++i)
    if (upath_size<W>(ws.paths[i]) < longest)
    {
      bits_set<W>(drop, i);
      ++n_drop;
    }
  return compact_paths<W>(ws, n_paths, drop, n_drop);
}

template <class W>
GTX_DEV uint32_t longest_of(AlignWorkspace const & ws, uint32_t n_paths) // :858-864
{
  uint32_t m = 0;
  if constexpr (DENSE_PP_KEYS)
  {
    for (uint32_t base = 0; base < n_paths; base += 64)
    {
      typename W::template PerLane<uint32_t> size;
      W::lanes([&](uint32_t l) { size[l] = base + l < n_paths ? path_size(ws.paths[base + l]) : 0u; });
      uint32_t const s = W::max(size);
      if (s > m)
        m = s;
    }
    return m;
  }
  for (uint32_t i = 0; i < n_paths; ++i)
  {
    uint32_t const s = upath_size<W>(ws.paths[i]);
    if (s > m)
      m = s;
  }
  return m;
}

template <class W>
GTX_DEV uint32_t remove_paths_with_too_many_mismatches(AlignWorkspace & ws, uint32_t n_paths) // :360-380
{
  if (n_paths == 0)
    return 0;
  uint32_t mn = 10;
  if constexpr (DENSE_PP_KEYS)
  {
    for (uint32_t base = 0; base < n_paths; base += 64)
    {
      typename W::template PerLane<uint32_t> inv; // (10 - mismatches, so that the lanes' maximum is the fewest mismatches)
      W::lanes([&](uint32_t l) {
        uint32_t const mm = base + l < n_paths ? static_cast<uint32_t>(ws.paths[base + l].mism) : 10u;
        inv[l] = mm < 10u ? 10u - mm : 0u;
      });
      uint32_t const best = 10u - W::max(inv);
      if (best < mn)
        mn = best;
    }
    uint32_t n_drop;
    auto const drop = mark_paths<W>(ws, n_paths, n_drop, [&](DPath const & p) { return static_cast<uint32_t>(p.mism) > mn; });
    return compact_paths<W>(ws, n_paths, drop, n_drop);
  }
  for (uint32_t i = 0; i < n_paths; ++i)
  {
    uint32_t const m = GTX_U(static_cast<uint32_t>(ws.paths[i].mism));
    if (m < mn)
      mn = m;
  }
  auto drop = new_path_set<W>(ws, n_paths);
  uint32_t n_drop = 0;
  for (uint32_t i = 0; i < n_paths; ++i)
    if (GTX_U(static_cast<uint32_t>(ws.paths[i].mism)) > mn)
    {
      bits_set<W>(drop, i);
      ++n_drop;
    }
  return compact_paths<W>(ws, n_paths, drop, n_drop);
}

template <class W, class Paths>
GTX_DEV bool all_paths_unique(Here, GraphView const & g, Paths const & paths, uint32_t n) // :219-231
{
  if (n < 2)
    return true;
  uint32_t const s0 = ug_ref_reach_pos<W>(g, GTX_U(paths[0].start)), e0 = ug_ref_reach_pos<W>(g, GTX_U(paths[0].end));
  if constexpr (DENSE_PP_KEYS)
  {
    for (uint32_t base = 1; base < n; base += 64)
    {
      typename W::template PerLane<bool> other;
      W::lanes([&](uint32_t l) {
        uint32_t const i = base + l;
        other[l] = i < n && s0 != g_ref_reach_pos(g, paths[i].start) && e0 != g_ref_reach_pos(g, paths[i].end);
      });
      if (W::ballot(other) != 0)
        return false;
    }
    return true;
  }
  for (uint32_t i = 1; i < n; ++i)
    if (s0 != ug_ref_reach_pos<W>(g, GTX_U(paths[i].start)) && e0 != ug_ref_reach_pos<W>(g, GTX_U(paths[i].end)))
      return false;
  return true;
}

template <class W>
GTX_DEV bool path_is_reference(DPath const & p) // path.cpp:176-185
{
  uint32_t const nv = GTX_U(static_cast<uint32_t>(p.nvar));
  for (uint32_t k = 0; k < nv; ++k)
    if (!(GTX_U(p.v[k].m[0]) & 1u))
      return false;
  return true;
}

template <class W>
GTX_DEV uint32_t remove_non_ref_paths_when_read_matches_ref(GraphView const & g, AlignWorkspace & ws, uint32_t n_paths) // :460-474
{
  if (all_paths_unique<W>(Here{}, g, ws.paths, n_paths))
    return n_paths;
  auto nonref = new_path_set<W>(ws, n_paths);
  uint32_t n_nonref = 0;
  for (uint32_t i = 0; i < n_paths; ++i)
    if (!path_is_reference<W>(ws.paths[i]))
    {
      bits_set<W>(nonref, i);
      ++n_nonref;
    }
  if (n_nonref == n_paths)
    return n_paths; // no path supports only the reference
  return compact_paths<W>(ws, n_paths, nonref, n_nonref);
}

template <class W>
GTX_DEV uint32_t remove_fully_special_paths(GraphView const & g, AlignWorkspace & ws, uint32_t n_paths) // :476-481
{
  auto drop = new_path_set<W>(ws, n_paths);
  uint32_t n_drop = 0;
  for (uint32_t i = 0; i < n_paths; ++i)
    if (ug_ref_reach_pos<W>(g, GTX_U(ws.paths[i].start)) == ug_ref_reach_pos<W>(g, GTX_U(ws.paths[i].end)))
    {
      bits_set<W>(drop, i);
      ++n_drop;
    }
  return compact_paths<W>(ws, n_paths, drop, n_drop);
}

template <class W>
GTX_DEV void remove_support_from_read_ends(GraphView const & g, AlignWorkspace & ws, uint32_t n_paths) // :382-432
{
  for (uint32_t i = 0; i < n_paths; ++i)
  {
    DPath & p = ws.paths[i];
    uint32_t const nvar = GTX_U(static_cast<uint32_t>(p.nvar));
    if (nvar == 0)
      continue;
    uint32_t const pstart = GTX_U(p.start), pend = GTX_U(p.end);
    bool const ss = g_is_special(g, pstart), es = g_is_special(g, pend);
    if (!ss && !es)
      continue;
    // std::minmax_element: first smallest, last largest
    uint32_t imin = 0, imax = 0;
    uint32_t omin = ug_site_order<W>(g, GTX_U(p.v[0].site)), omax = omin;
    for (uint32_t k = 1; k < nvar; ++k)
    {
      uint32_t const o = ug_site_order<W>(g, GTX_U(p.v[k].site));
      if (o < omin)
      {
        omin = o;
        imin = k;
      }
      if (!(o < omax))
      {
        omax = o;
        imax = k;
      }
    }
    bool const clear_max = es && static_cast<int64_t>(ug_actual_pos<W>(g, pend)) <= static_cast<int64_t>(omax) + 4;
    bool clear_min = false;
    if (ss)
    {
      bool ambiguous = true;
      if (g_is_special(g, pstart + 4u))
        ambiguous = ug_ref_reach_pos<W>(g, pstart) != ug_ref_reach_pos<W>(g, pstart + 4u);
      clear_min = ambiguous;
    }
    GTX_LEAD
    {
      if (clear_max)
        pv_clear(p.v[imax]);
      if (clear_min)
        pv_clear(p.v[imin]);
    }
    W::lds_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GenotypePaths::walk_read_ends / walk_read_starts (genotype_paths.cpp:483-553 / 555-621)
// ---------------------------------------------------------------------------------------------------------------
template <class W>
GTX_DEV void walk_read(GraphView const & g, AlignWorkspace & ws, bool starts, uint32_t & n_paths, uint32_t & longest,
                       uint32_t & status)
{
  uint32_t const L = GTX_U(ws.read_len);
  if (n_paths == 0 || upath_size<W>(ws.paths[0]) == L)
    return;
  if (n_paths > MAX_SEED_NUMBER_FOR_WALKING)
    return;
  int maximum_mismatches = -1;
  if (n_paths > MAX_SEED_NUMBER_ALLOWING_MISMATCHES)
    maximum_mismatches = 0;
  uint32_t best = 7;
  uint32_t n_wl = 0, n_wlists = 0;
  WalkBuffers & wb = ws.u.w;
  // Paths that end at the same ordinary position with the same part of the read left over (the +1-mismatch copies of a
  // chain, the other alleles of a site behind them) ask for the same walk: its outcome depends on the path only through
  // the variant nodes get_locations offers, and an ordinary position inside a reference node offers none.  The labels
  // of the last such walk stay in wb.dfs_out; a second identical request reuses them (the budget can only have shrunk
  // to that walk's own mismatch count or below: same labels or none).
  uint32_t memo_anchor = INVALID, memo_idx = 0, memo_nl = 0, memo_mm = 0;
  GTX_PROF_WBEGIN
  for (uint32_t i = 0; i < n_paths; ++i)
  {
    DPath const & path = ws.paths[i];
    uint32_t const prs = GTX_U(static_cast<uint32_t>(path.rs)), pre = GTX_U(static_cast<uint32_t>(path.re));
    SubRead sr;
    sr.rd = ws.rd;
    if (starts)
    {
      if (prs == 0)
        continue;
      sr.begin = 0;
      sr.len = prs + 1u;
    }
    else
    {
      if (pre == L - 1)
        continue;
      sr.begin = pre;
      sr.len = L - pre;
    }
    uint32_t mm;
    if (maximum_mismatches < 0)
    {
      uint32_t const budget = 2 + sr.len / 11;
      mm = budget < best ? budget : best;
    }
    else
      mm = static_cast<uint32_t>(maximum_mismatches);
    uint32_t const anchor = GTX_U(starts ? path.start : path.end);
    uint32_t nl;
    bool const reuse = anchor == memo_anchor && memo_idx == (starts ? prs : pre);
    // Shortcut for the common geometry: the anchor is an ordinary position inside a reference node and the sub-read
    // fits in what is left of that node.  get_locations would return that single 'R' location and the walk a single
    // sequence without ever reaching a variant site (graph.cpp:1232-1243 / 1484-1496), so the result is one id-less
    // label or nothing -- computed here without going through the location / candidate tables.
    bool shortcut = reuse;
    if (reuse)
    {
      nl = (memo_nl != 0 && memo_mm <= mm) ? memo_nl : 0;
      mm = nl ? memo_mm : mm;
    }
    if (!shortcut && !starts && g.pos_info && !g_is_special(g, anchor) && anchor >= g.first_order && g.n_ref > 1 && anchor - g.first_order < g.n_pos_info &&
        sr.len <= 255)
    {
      // the same shortcut through the position table: one lookup instead of bucket -> node order -> node tables
      uint32_t const w = GTX_U(g.pos_info[anchor - g.first_order]);
      if (w != INVALID && (w & 255u) >= sr.len)
      {
        shortcut = true;
        uint8_t const * dna = reinterpret_cast<uint8_t const *>(g.dna) + (w >> 8);
        uint32_t const budget = mm;
        uint32_t const got = cmp_codes<W, false>(sr, 0, dna, w & 255u, 0, budget);
        if (got <= budget)
        {
          mm = got;
          nl = 1;
          GTX_LEAD wb.dfs_out[0] = DevLabel{anchor, anchor + (sr.len - 1), INVALID, 0};
          W::lds_sync();
        }
        else
          nl = 0;
      }
    }
    if (!shortcut && starts && g.pos_info && g.pos_back && !g_is_special(g, anchor) && anchor >= g.first_order && g.n_ref > 1 &&
        anchor - g.first_order < g.n_pos_info && sr.len <= 255)
    {
      // ... and backwards: the node has to reach far enough in front of the anchor (pos_back, capped at 255)
      uint32_t const w = GTX_U(g.pos_info[anchor - g.first_order]);
      uint32_t const back = GTX_U(static_cast<uint32_t>(g.pos_back[anchor - g.first_order]));
      if (w != INVALID && back >= sr.len - 1)
      {
        shortcut = true;
        uint8_t const * dna = reinterpret_cast<uint8_t const *>(g.dna) + (w >> 8) - (sr.len - 1); // (its last character is the anchor's)
        uint32_t const budget = mm;
        uint32_t const got = cmp_codes<W, true>(sr, 0, dna, sr.len, 0, budget);
        if (got <= budget)
        {
          mm = got;
          nl = 1;
          GTX_LEAD wb.dfs_out[0] = DevLabel{anchor - (sr.len - 1), anchor, INVALID, 0};
          W::lds_sync();
        }
        else
          nl = 0;
      }
    }
    if (!shortcut && !g_is_special(g, anchor) && anchor >= g.first_order && g.n_ref > 1)
    {
      uint32_t const rr = g_ref_node_at<W>(g, anchor);
      uint32_t const ro = GTX_U(g.ref_order[rr]), rl = GTX_U(g.ref_len[rr]);
      if (anchor < ro + rl)
      {
        uint32_t const offset = anchor - ro;
        uint32_t const avail = starts ? offset + 1 : rl - offset;
        if (avail >= sr.len)
        {
          shortcut = true;
          uint8_t const * dna = reinterpret_cast<uint8_t const *>(g.dna) + GTX_U(g.ref_dna[rr]);
          uint32_t const budget = mm;
          uint32_t const got = starts ? cmp_codes<W, true>(sr, 0, dna, offset + 1, 0, budget)
                                      : cmp_codes<W, false>(sr, 0, dna + offset, avail, 0, budget);
          if (got <= budget)
          {
            mm = got;
            nl = 1;
            GTX_LEAD wb.dfs_out[0] = starts ? DevLabel{anchor - (sr.len - 1), anchor, INVALID, 0}
                                            : DevLabel{anchor, anchor + (sr.len - 1), INVALID, 0};
            W::lds_sync();
          }
          else
            nl = 0;
        }
      }
    }
    GTX_PROF_WTICK(2)
    if (!shortcut)
    {
      uint32_t const n_locs = get_locations<W>(g, anchor, path, wb.locs, AlignCfg::LOC_CAP, status);
      W::lds_sync();
      GTX_PROF_WTICK(3)
      if (status)
        return;
      if (n_locs == 0 || n_locs > MAX_NUM_LOCATIONS_PER_PATH)
        continue;
      nl = iterative_dfs<W>(g, ws, n_locs, starts, sr, mm, status);
      GTX_PROF_WTICK(4)
      if (status)
        return;
      // remember the walk when its start did not depend on the path: one location, inside a reference node
      bool const plain = n_locs == 1 && GTX_U(wb.locs[0].type) != 2 && g.pos_info && anchor >= g.first_order &&
                         anchor - g.first_order < g.n_pos_info && GTX_U(g.pos_info[anchor - g.first_order]) != INVALID;
      memo_anchor = plain ? anchor : INVALID;
      memo_idx = starts ? prs : pre;
      memo_nl = nl;
      memo_mm = mm;
    }
    else if (!reuse)
      memo_anchor = INVALID; // (the shortcut wrote its label to wb.dfs_out)
    if (nl == 0)
      continue;
    if (mm < best)
    {
      n_wl = 0;
      n_wlists = 0;
      best = mm;
    }
    if (mm == best)
    {
      if (n_wlists >= AlignCfg::WLISTS || n_wl + nl > cap_wl(ws))
      {
        status |= GTX_ST_DFS_OVERFLOW;
        return;
      }
      for (uint32_t k = 0; k < nl; ++k)
        copy_entry<W>(ws.wl[n_wl + k], wb.dfs_out[k]);
      GTX_LEAD
      {
        ws.wl_idx[n_wlists] = starts ? prs : pre;
        ws.wl_off[n_wlists] = n_wl;
        ws.wl_off[n_wlists + 1] = n_wl + nl;
      }
      W::lds_sync();
      n_wl += nl;
      ++n_wlists;
    }
  }
  for (uint32_t k = 0; k < n_wlists; ++k)
  {
    uint32_t const o0 = GTX_U(ws.wl_off[k]);
    DevLabel const * ll = ws.wl + o0;
    uint32_t const n = GTX_U(ws.wl_off[k + 1]) - o0;
    uint32_t const idx = GTX_U(ws.wl_idx[k]);
    if (starts)
      add_kmer_labels<W>(ws, ll, n, 0, idx, best, true, n_paths, longest, status);
    else
      add_kmer_labels<W>(ws, ll, n, idx, L - 1, best, false, n_paths, longest, status);
    if (status)
      return;
  }
  GTX_PROF_WTICK(5)
}

// ---------------------------------------------------------------------------------------------------------------
// k-mer keys (src/utilities/type_conversions.cpp:207-288) and index probes (src/index/ph_index.cpp:66-107)
// ---------------------------------------------------------------------------------------------------------------

// to_uint64_vec for a k-mer with ambiguous bases; sequential by nature (list order is the contract), leader only.
// Returns the number of keys (0 = gave up, > 97 partial keys)
// (keys are produced in plane form: appending base value b at base index t sets bit t of the low / high word)
GTX_DEV uint32_t expand_keys(uint8_t const * rd, uint32_t at, uint64_t * keys, uint32_t cap)
{
  uint32_t n = 1;
  keys[0] = 0;
  for (uint32_t t = 0; t < K; ++t)
  {
    uint32_t const origin = n;
    if (origin > 97)
      return 0;
    uint32_t const code = rd[at + t] & 15u;
    uint32_t const fan = (code == 15u || code == 0u) ? 4u : static_cast<uint32_t>(__builtin_popcount(code));
    if (fan * origin > cap)
      return 0xFFFFFFFFu; // the list outgrows this pass' key buffer
    auto with = [t](uint64_t k, uint64_t b) { return k | ((b & 1u) << t) | ((b >> 1) << (32 + t)); };
    for (uint32_t u = 0; u < origin; ++u)
    {
      if (code == 15u || code == 0u)
      {
        keys[n++] = with(keys[u], 0);
        keys[n++] = with(keys[u], 1);
        keys[n++] = with(keys[u], 2);
        keys[u] = with(keys[u], 3);
      }
      else
      {
        int left = __builtin_popcount(code);
        uint64_t const base = keys[u];
        for (uint32_t b = 0; b < 4; ++b)
        {
          if (!(code & (1u << b)))
            continue;
          if (left == 1)
            keys[u] = with(base, b);
          else
            keys[n++] = with(base, b);
          --left;
        }
      }
    }
  }
  return n;
}

// The same list, a key per lane, for any number of ambiguous bases (`amb`: their places in the k-mer; `base`: the key of the
// other bases).  The list's order is a mixed-radix number read from the last ambiguous base backwards: when base j (set S_j of
// s_j bases) meets a list of n entries, entry u keeps its place with the LAST base of S_j and entries n + u (s_j - 1) + k are its
// copies with the k-th of the other bases, ascending -- so entry e of the final list is decoded by peeling one base at a time:
// e < n: last(S_j), parent e; else parent (e - n) / (s_j - 1), base (e - n) % (s_j - 1).  The reference gives up (empty list)
// when it starts ANY base position with more than 97 keys, so a list may only outgrow 97 on the k-mer's last base.
// Returns the number of keys in ws.u.keybuf, 0 = gave up, 0xFFFFFFFF = the list outgrows this pass' key buffer.
// (the leader's loop of expand_keys -- 32 positions times every key, each a dependent LDS update -- was 0.4 M cycles for three Ns)
template <class W, class WS>
GTX_DEV uint32_t expand_keys_lanes(WS & ws, uint32_t at, uint32_t amb, uint64_t base)
{
  auto set_of = [&](uint32_t t) {
    uint32_t const code = GTX_U(static_cast<uint32_t>(ws.rd[at + t])) & 15u;
    return (code == 0u || code == 15u) ? 15u : code;
  };
  uint32_t n = 1;
  for (uint32_t rest = amb; rest; rest &= rest - 1u)
  {
    uint32_t const t = static_cast<uint32_t>(__builtin_ctz(rest));
    (void)t;
    if (n > 97)
      return 0; // (the position behind the one that made the list this long)
    uint32_t const fan = static_cast<uint32_t>(__builtin_popcount(set_of(t)));
    if (fan * n > AlignCfg::KEY_CAP)
      return 0xFFFFFFFFu;
    n *= fan;
  }
  if (n > 97 && (amb >> 31) == 0u)
    return 0; // (a base position follows the last ambiguous one)
  uint32_t const total = n;
  for (uint32_t b = 0; b < total; b += 64)
    W::lanes([&](uint32_t l) {
      uint32_t e = b + l;
      if (e < total)
      {
        uint64_t key = base;
        uint32_t size = total;
        for (uint32_t rest = amb; rest;)
        {
          uint32_t const t = 31u - static_cast<uint32_t>(__builtin_clz(rest));
          rest &= ~(1u << t);
          uint32_t const set = set_of(t), fan = static_cast<uint32_t>(__builtin_popcount(set));
          uint32_t const last = 31u - static_cast<uint32_t>(__builtin_clz(set));
          uint32_t const before = size / fan;
          uint32_t bs = last;
          if (e >= before)
          {
            uint32_t const k = (e - before) % (fan - 1u);
            e = (e - before) / (fan - 1u);
            uint32_t others = set & ~(1u << last);
            for (uint32_t x = 0; x < k; ++x)
              others &= others - 1u;
            bs = static_cast<uint32_t>(__builtin_ctz(others));
          }
          key |= (static_cast<uint64_t>(bs & 1u) << t) | (static_cast<uint64_t>(bs >> 1) << (32u + t));
          size = before;
        }
        ws.u.keybuf[b + l] = key;
      }
    });
  W::lds_sync();
  return total;
}

// Wave-parallel probe of a key list (`nkeys` keys: either keybuf[0..nkeys) or the 96 Hamming-1 neighbours of `base`
// generated on the fly) with the multi_get rule: a list of more than one key whose hits total more than
// max_index_labels yields nothing.  Labels land in ws.lbl in key order, bucket order inside a key (stable prefix-sum
// compaction).  Returns the number of labels.
template <class W>
GTX_DEV uint32_t probe_list(IndexView const & ix, AlignWorkspace & ws, bool hamming, uint64_t base, uint32_t nkeys, uint32_t & status)
{
  uint32_t const rounds = (nkeys + 63) / 64;
  uint64_t * kres = ws.u.keybuf; // key j is replaced by (offset | count << 32)
  uint32_t total = 0;
  for (uint32_t t = 0; t < rounds; ++t)
  {
    typename W::template PerLane<uint32_t> cnt;
    W::lanes([&](uint32_t l) {
      uint32_t const j = t * 64 + l;
      uint32_t o = 0, c = 0;
      if (j < nkeys)
      {
        // neighbour j of the reference's list (type_conversions.cpp:272-288): base 31 - j/3 changed by xor j%3 + 1
        uint32_t const hb = 31u - j / 3u, hm = j % 3u + 1u;
        uint64_t const key = hamming ? base ^ ((static_cast<uint64_t>(hm & 1u) << hb) | (static_cast<uint64_t>(hm >> 1) << (32u + hb)))
                                     : kres[j];
        index_find(ix, key, o, c);
        kres[j] = static_cast<uint64_t>(o) | (static_cast<uint64_t>(c) << 32);
      }
      cnt[l] = c;
    });
    total += W::sum(cnt);
  }
  total = GTX_U(total);
  W::lds_sync();
  if (nkeys > 1 && total > ix.max_index_labels)
    return 0; // ph_index.cpp:84-89
  if (total > cap_lbl(ws))
  {
    status |= GTX_ST_LABEL_OVERFLOW;
    return 0;
  }
  if (total == 0)
    return 0;
  uint32_t done = 0;
  for (uint32_t t = 0; t < rounds; ++t)
  {
    typename W::template PerLane<uint32_t> cnt, pre;
    W::lanes([&](uint32_t l) {
      uint32_t const j = t * 64 + l;
      cnt[l] = j < nkeys ? static_cast<uint32_t>(kres[j] >> 32) : 0u;
    });
    uint32_t round_total;
    W::excl_scan(cnt, pre, round_total);
    if (round_total != 0)
      W::lanes([&](uint32_t l) {
        uint32_t const j = t * 64 + l;
        uint32_t const c = cnt[l];
        if (c != 0)
        {
          uint32_t const o = static_cast<uint32_t>(kres[j]);
          for (uint32_t k = 0; k < c; ++k)
            ws.lbl[done + pre[l] + k] = ix.labels[o + k];
        }
      });
    done += round_total;
  }
  W::lds_sync();
  return total;
}

// Orders the collected candidates ((label offset) | (label count << 32) | (j << 56) in ws.u.keybuf) by neighbour number,
// applies the >max_index_labels rule of multi_get (ph_index.cpp:84-89; the list has 96 keys) and copies their labels to
// ws.lbl.  Returns the number of labels.
template <class W>
GTX_DEV uint32_t hamming1_finish(IndexView const & ix, AlignWorkspace & ws, uint32_t ncand)
{
  uint64_t * cand = ws.u.keybuf;
  if (ncand == 0)
    return 0;
  if (ncand > 1)
  {
    GTX_LEAD
    {
      for (uint32_t a = 1; a < ncand; ++a)
      {
        uint64_t const x = cand[a];
        uint32_t b = a;
        while (b > 0 && (cand[b - 1] >> 56) > (x >> 56))
        {
          cand[b] = cand[b - 1];
          --b;
        }
        cand[b] = x;
      }
    }
    W::lds_sync();
  }
  uint32_t total = 0;
  for (uint32_t a = 0; a < ncand; ++a)
    total += GTX_U(static_cast<uint32_t>(cand[a] >> 32)) & 0xFFFFFFu;
  if (total > ix.max_index_labels)
    return 0;
  // (a list the reference keeps -- 75 labels or fewer -- that this pass' table does not hold: 0xFFFFFFFF, the caller sends the task
  //  on.  Round 5: seven or eight SNP sites under one k-mer give its 96 neighbours 49 / 64 labels; they were written past the
  //  table's 40 entries into what lies behind it in the workspace, and the task's record was nonsense without a status.)
  if (total > cap_lbl(ws))
    return 0xFFFFFFFFu;
  uint32_t done = 0;
  for (uint32_t a = 0; a < ncand; ++a)
  {
    uint64_t const e = GTX_U(cand[a]);
    uint32_t const off = static_cast<uint32_t>(e), cnt = static_cast<uint32_t>(e >> 32) & 0xFFFFFFu;
    for (uint32_t b = 0; b < cnt; b += 64)
      W::lanes([&](uint32_t l) {
        if (b + l < cnt)
          ws.lbl[done + b + l] = ix.labels[off + b + l];
      });
    done += cnt;
  }
  W::lds_sync();
  return total;
}

// The Hamming-1 list of a unique exact key `q` (kmer_help_functions.cpp:97-119 + ph_index.cpp:66-107) without probing
// the 96 neighbours: an indexed key at Hamming distance 1 differs from q in one base, so it agrees with q on the left
// 16 bases or on the right 16 bases -- two bucket lookups find every candidate.  Candidates are put in the order the
// reference visits them (neighbour j = 3*bb + m-1) before their labels are copied.  Returns false when a bucket is too
// large for this route (the caller then probes the 96 keys directly); n_lbl is the number of labels placed in ws.lbl.
template <class W>
GTX_DEV bool hamming1_by_halves(IndexView const & ix, AlignWorkspace & ws, uint64_t q, uint32_t & n_lbl)
{
  uint32_t o[2], c[2];
  half_find(ix, half_key(q, 0), o[0], c[0]);
  half_find(ix, half_key(q, 1), o[1], c[1]);
  if (c[0] > ix.half_bucket_cap || c[1] > ix.half_bucket_cap)
    return false;
  uint64_t * cand = ws.u.keybuf;
  uint32_t ncand = 0;
  for (uint32_t side = 0; side < 2; ++side)
  {
    uint32_t const cs = c[side], os = o[side];
    if (cs == 0)
      continue;
    typename W::template PerLane<uint32_t> valid, pre;
    typename W::template PerLane<uint64_t> packed;
    W::lanes([&](uint32_t l) {
      uint32_t ok = 0, j = 0;
      uint64_t pk = 0;
      if (l < cs)
      {
        HalfEntry const e = ix.hlist[os + l];
        if (hamming1_neighbour(e.key, q, j))
        {
          ok = 1;
          pk = static_cast<uint64_t>(e.off) | (static_cast<uint64_t>(e.cnt) << 32) | (static_cast<uint64_t>(j) << 56);
        }
      }
      valid[l] = ok;
      packed[l] = pk;
    });
    uint32_t nv;
    W::excl_scan(valid, pre, nv);
    if (nv != 0)
      W::lanes([&](uint32_t l) {
        if (valid[l])
          cand[ncand + pre[l]] = packed[l];
      });
    ncand += nv;
  }
  W::lds_sync();
  n_lbl = hamming1_finish<W>(ix, ws, ncand);
  return true;
}

// Same list from the bucket entries that were fetched up front (ws.he): at most 2*HE_CAP entries, wave-uniform code
template <class W>
GTX_DEV uint32_t hamming1_from_cache(IndexView const & ix, AlignWorkspace & ws, uint32_t i, uint64_t q)
{
  uint64_t * cand = ws.u.keybuf;
  uint32_t ncand = 0;
  for (uint32_t side = 0; side < 2; ++side)
  {
    uint32_t const cs = GTX_U(ws.hcnt[i][side]);
    for (uint32_t e = 0; e < cs; ++e)
    {
      uint32_t j;
      if (hamming1_neighbour(GTX_U(ws.he[i][side][e].key), q, j))
      {
        uint64_t const pk = static_cast<uint64_t>(ws.he[i][side][e].off) | (static_cast<uint64_t>(ws.he[i][side][e].cnt) << 32) |
                            (static_cast<uint64_t>(j) << 56);
        GTX_LEAD cand[ncand] = pk;
        ++ncand;
      }
    }
  }
  if (ncand == 0)
    return 0;
  W::lds_sync();
  return hamming1_finish<W>(ix, ws, ncand);
}

// Everything after seeding for the read the fast seeding produced: one path without variant sites that starts at read
// base 0.  walk_read_starts has nothing to do; walk_read_ends extends it iff the rest of the read matches the reference
// node the path ends in within the budget (the shortcut of walk_read); of the filters only "more than 10 mismatches"
// can act on a single path.  Returns false when the geometry needs the general code (path end not inside a reference
// node with room for the rest of the read).
template <class W, class WS>
GTX_DEV bool finish_single_path(Here, GraphView const & g, WS & ws, uint32_t & n_paths, uint32_t & longest)
{
  uint32_t const L = GTX_U(ws.read_len);
  DPath & p = ws.paths[0];
  uint32_t const pre = GTX_U(static_cast<uint32_t>(p.re));
  uint32_t mism = GTX_U(static_cast<uint32_t>(p.mism));
  if (pre != L - 1)
  {
    uint32_t const anchor = GTX_U(p.end);
    if (g_is_special(g, anchor) || anchor < g.first_order || g.n_ref <= 1)
      return false;
    SubRead sr;
    sr.rd = ws.rd;
    sr.begin = pre;
    sr.len = L - pre;
    uint8_t const * dna;
    uint32_t room;
    if (g.pos_info)
    {
      // one table lookup instead of the bucket -> node order -> node tables chain (this wave is latency-bound)
      if (anchor - g.first_order >= g.n_pos_info)
        return false;
      uint32_t const w = GTX_U(g.pos_info[anchor - g.first_order]);
      if (w == INVALID || (w & 255u) < sr.len) // (room is capped at 255: a longer tail takes the general code)
        return false;
      room = w & 255u;
      dna = reinterpret_cast<uint8_t const *>(g.dna) + (w >> 8);
    }
    else
    {
      uint32_t const rr = g_ref_node_at<W>(g, anchor);
      uint32_t const ro = GTX_U(g.ref_order[rr]), rl = GTX_U(g.ref_len[rr]);
      if (!(anchor < ro + rl) || rl - (anchor - ro) < sr.len)
        return false;
      room = rl - (anchor - ro);
      dna = reinterpret_cast<uint8_t const *>(g.dna) + GTX_U(g.ref_dna[rr]) + (anchor - ro);
    }
    uint32_t const budget = 2 + sr.len / 11 < 7 ? 2 + sr.len / 11 : 7; // genotype_paths.cpp:505-511, best starts at 7
    uint32_t const got = cmp_codes<W, false>(sr, 0, dna, room, 0, budget);
    if (got <= budget)
    {
      mism += got;
      GTX_LEAD
      {
        p.end = anchor + (sr.len - 1);
        p.re = static_cast<uint16_t>(L - 1);
        p.mism = static_cast<uint16_t>(mism);
      }
      W::lds_sync();
      longest = L;
    }
  }
  if (mism > 10) // remove_paths_with_too_many_mismatches (genotype_paths.cpp:360-380) on one path
  {
    n_paths = 0;
    longest = 0;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// one (read, orientation): find_genotype_paths_of_one_of_the_sequences (alignment.cpp:23-103)
// ---------------------------------------------------------------------------------------------------------------
// Stages shared by both passes: unpack the read, build the exact keys, issue every index lookup of the read at once, stage
// what they point at, and (try_fast) attempt the fast seeding.  Returns true when the fast seeding produced the read's
// single path in ws.paths[0] (n_paths = 1, longest set).
template <class W, class WS>
GTX_DEV bool seed_stage(Here, GraphView const & g, IndexView const & ix, WS & ws, uint8_t const * seq4, uint32_t len, bool reverse,
                        bool try_fast, uint32_t & n_paths, uint32_t & longest)
{
  GTX_PROF_BEGIN
  // -- load the read: bit planes (graph_dev.hpp; seq4 = the read's row) -> one code per byte, four bases per lane and one
  //    4-byte store: reads up to 256 bp in one pass.  The reverse orientation is the reverse complement, and complementing
  //    an IUPAC code is reversing its 4 bits (A<->T, C<->G): the forward codes go to LDS first and are turned around there.
  static_assert(AlignCfg::MAX_READ <= 256 && AlignCfg::MAX_READ % 4 == 0, "one pass of 64 lanes x 4 bases");
  W::lanes([&](uint32_t l) {
    if (4 * l < len)
    {
      uint32_t const * gq = reinterpret_cast<uint32_t const *>(seq4) + 4u * ((4 * l) >> 5);
      uint32_t packed = plane_codes4(gq[0], gq[1], gq[2], gq[3], (4 * l) & 31u);
      // bases behind the read's end count as N, and so does '=' (assigned to a seqan Iupac it becomes N,
      // hts_parallel_reader.cpp:226-243): both are zero bytes by now
      uint32_t const inside = len - 4 * l >= 4 ? 0xFFFFFFFFu : (1u << (8 * (len - 4 * l))) - 1u;
      packed &= inside;
      uint32_t const zero = ~(packed | (packed >> 1) | (packed >> 2) | (packed >> 3)) & 0x01010101u;
      packed |= zero * 15u;
      reinterpret_cast<uint32_t *>(ws.rd)[l] = packed;
    }
  });
  if (reverse)
  {
    W::lds_sync();
    typename W::template PerLane<uint32_t> turned;
    W::lanes([&](uint32_t l) {
      uint32_t packed = 0;
      for (uint32_t k = 0; k < 4; ++k)
      {
        uint32_t const i = 4 * l + k;
        uint32_t c = 15;
        if (i < len)
        {
          c = ws.rd[len - 1 - i];
          c = ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3);
        }
        packed |= c << (8 * k);
      }
      turned[l] = packed;
    });
    W::lds_sync();
    W::lanes([&](uint32_t l) {
      if (4 * l < len)
        reinterpret_cast<uint32_t *>(ws.rd)[l] = turned[l];
    });
  }
  GTX_LEAD ws.read_len = len;
  W::lds_sync();
  GTX_PROF_TICK(0)

  uint32_t const n_k = len < K ? 0 : 1 + (len - K) / (K - 1); // kmer_help_functions.cpp:10-17
  // -- exact keys of every k-mer.  Unambiguous k-mer: lanes 0..31 each hold one base, two ballots give the low/high
  //    bit planes, interleaving them gives the key (first base in the top bits, type_conversions.cpp:75-87).
  //    Two k-mers per pass: lanes 0..31 hold k-mer i, lanes 32..63 k-mer i+1; each ballot serves both.
  static_assert(K == 32, "two 32-base k-mers per 64-lane wave");
  for (uint32_t i = 0; i < n_k; i += 2)
  {
    typename W::template PerLane<bool> amb_l, b0_l, b1_l;
    W::lanes([&](uint32_t l) {
      uint32_t const km = i + (l >> 5);
      uint32_t const c = km < n_k ? ws.rd[(K - 1) * km + (l & 31u)] : 1u;
      bool const single = (c & (c - 1u)) == 0u && c != 0u;
      uint32_t const two = (c == 2u) ? 1u : (c == 4u) ? 2u : (c == 8u) ? 3u : 0u;
      amb_l[l] = !single;
      b0_l[l] = (two & 1u) != 0u;
      b1_l[l] = (two & 2u) != 0u;
    });
    uint64_t const amb = W::ballot(amb_l), b0 = W::ballot(b0_l), b1 = W::ballot(b1_l);
    GTX_LEAD
    {
      for (uint32_t h = 0; h < 2 && i + h < n_k; ++h)
      {
        uint32_t const a = static_cast<uint32_t>(amb >> (32 * h));
        // plane form (gtx_flat.hpp: plane_key): low word = low bits of the bases, high word = high bits
        ws.key0[i + h] = (static_cast<uint64_t>(static_cast<uint32_t>(b1 >> (32 * h))) << 32) | static_cast<uint32_t>(b0 >> (32 * h));
        ws.nkeys0[i + h] = a == 0 ? 1 : 2; // 2 = "not a single key"; that list is generated when the k-mer is processed
        ws.cnt0[i + h] = 0;
        ws.off0[i + h] = a; // (multi-key k-mers have no exact slot: keep the ambiguous positions here)
      }
    }
  }
  W::lds_sync();
  // -- all index lookups of the read at once, one per lane, so that their memory latencies overlap:
  //    lane 3i: exact key of k-mer i (PHIndex::get), lanes 3i+1 / 3i+2: its left / right half-key bucket;
  //    lanes 32 + 4i + w: key w of k-mer i when it has exactly one ambiguous base (its list has 2..4 keys)
  uint32_t const kc = n_k < AlignCfg::KC ? n_k : AlignCfg::KC;
  bool const use_halves = ix.half_bucket_cap != 0;
  static_assert(3 * AlignCfg::MAX_KMERS <= 32 && 32 + 4 * AlignCfg::MAX_KMERS <= 64, "lane map of the lookups");
  W::lanes([&](uint32_t l) {
    uint32_t const i = l / 3, w = l % 3;
    if (l >= 32)
    {
      uint32_t const ai = (l - 32) >> 2, aw = (l - 32) & 3u;
      if (ai < n_k && ws.nkeys0[ai] != 1)
      {
        uint32_t const amb = ws.off0[ai];
        uint32_t off = 0, cnt = aw == 0 ? 0xFFFFFFFFu : 0u; // several ambiguous bases: "unknown", left to the general loop
        if ((amb & (amb - 1u)) == 0u)
        {
          // list order of to_uint64_vec: the last admissible base first, then the others ascending (see the loop below)
          uint32_t const t0 = static_cast<uint32_t>(__builtin_ctz(amb));
          uint32_t const code = ws.rd[(K - 1) * ai + t0] & 15u;
          uint32_t const set = (code == 0u || code == 15u) ? 15u : code;
          cnt = 0;
          if (aw < static_cast<uint32_t>(__builtin_popcount(set)))
          {
            uint32_t const last = 31u - static_cast<uint32_t>(__builtin_clz(set));
            uint32_t b = last;
            if (aw > 0)
            {
              uint32_t rest = set & ~(1u << last);
              for (uint32_t k = 1; k < aw; ++k)
                rest &= rest - 1u;
              b = static_cast<uint32_t>(__builtin_ctz(rest));
            }
            uint64_t const key = ws.key0[ai] | (static_cast<uint64_t>(b & 1u) << t0) | (static_cast<uint64_t>(b >> 1) << (32u + t0));
            bucket_find(ix.slots, ix.log2_cap, key, off, cnt);
          }
        }
        ws.aoff[ai][aw] = off;
        ws.acnt[ai][aw] = cnt;
      }
    }
    else if (l < 3 * n_k && ws.nkeys0[i] == 1 && (w == 0 || (i < kc && use_halves)))
    {
      uint64_t const q = ws.key0[i];
      uint32_t off, cnt;
      IndexSlot const * hit;
      bucket_find(w == 0 ? ix.slots : ix.hslots, w == 0 ? ix.log2_cap : ix.h_log2_cap, w == 0 ? q : half_key(q, w - 1), off, cnt,
                  &hit);
      uint32_t * o = w == 0 ? &ws.off0[i] : &ws.hoff[i][w - 1];
      uint32_t * c = w == 0 ? &ws.cnt0[i] : &ws.hcnt[i][w - 1];
      *o = off;
      *c = cnt;
      if (cnt == 1 && i < kc)
      {
        // a single label / a single bucket entry is inline in the slot (same cache line): no second round trip
        uint4_t const payload = *reinterpret_cast<uint4_t const *>(hit->p);
        static_assert(sizeof(HalfEntry) == 16 && sizeof(DevLabel) == 16, "staged entries are 16-byte words");
        *(w == 0 ? reinterpret_cast<uint4_t *>(&ws.xl[i][0]) : reinterpret_cast<uint4_t *>(&ws.he[i][w - 1][0])) = payload;
      }
    }
  });
  W::lds_sync();
  // -- ... and everything those lookups point at that is small enough to be staged: bucket entries and exact labels
  W::lanes([&](uint32_t l) {
    // both kinds of staged entry are 16 bytes: one predicated 16-byte copy per lane
    constexpr uint32_t NH = 2 * AlignCfg::HE_CAP;
    bool const is_half = l < AlignCfg::KC * NH;
    uint32_t const m = is_half ? l : l - AlignCfg::KC * NH;
    uint32_t const per = is_half ? NH : AlignCfg::XL_CAP;
    uint32_t const i = m / per, e = is_half ? m % AlignCfg::HE_CAP : m % AlignCfg::XL_CAP;
    uint32_t const side = (m / AlignCfg::HE_CAP) % 2;
    if (l < AlignCfg::KC * (NH + AlignCfg::XL_CAP) && i < kc && ws.nkeys0[i] != 1)
    {
      // k-mer with one ambiguous base whose keys have exactly one label between them: stage that label
      if (!is_half && e == 0)
      {
        uint32_t const c0 = ws.acnt[i][0], c1 = ws.acnt[i][1], c2 = ws.acnt[i][2], c3 = ws.acnt[i][3];
        if (c0 <= 1 && c0 + c1 + c2 + c3 == 1) // (c0 = 0xFFFFFFFF marks "unknown")
        {
          uint32_t const w1 = c0 ? 0u : c1 ? 1u : c2 ? 2u : 3u;
          ws.xl[i][0] = ix.labels[ws.aoff[i][w1]];
          ws.cnt0[i] = 1; // (cnt0 of a multi-key k-mer is otherwise 0 and unused)
        }
      }
    }
    else if (l < AlignCfg::KC * (NH + AlignCfg::XL_CAP) && i < kc && (!is_half || use_halves))
    {
      uint32_t const cnt = is_half ? ws.hcnt[i][side] : ws.cnt0[i];
      uint32_t const off = is_half ? ws.hoff[i][side] : ws.off0[i];
      if (cnt > 1 && cnt <= (is_half ? AlignCfg::HE_CAP : AlignCfg::XL_CAP) && e < cnt) // (a single entry came with its slot)
      {
        static_assert(sizeof(HalfEntry) == 16 && sizeof(DevLabel) == 16, "staged entries are copied as 16-byte words");
        uint4_t const * src = is_half ? reinterpret_cast<uint4_t const *>(ix.hlist + off + e) : reinterpret_cast<uint4_t const *>(ix.labels + off + e);
        uint4_t * dst = is_half ? reinterpret_cast<uint4_t *>(&ws.he[i][side][e]) : reinterpret_cast<uint4_t *>(&ws.xl[i][e]);
        *dst = *src;
      }
    }
  });
  static_assert(AlignCfg::KC * (2 * AlignCfg::HE_CAP + AlignCfg::XL_CAP) <= 64, "staging needs one lane per entry");
  W::lds_sync();
  // -- fast seeding.  Most reads: every k-mer has exactly one label in its exact and Hamming-1 lists together (an exact
  //    hit without indexed neighbours, or one substitution error and a single neighbour), none of them on a variant, and
  //    consecutive labels abut.  The loop below would chain them into one path; that path is written directly, one
  //    k-mer per lane.  Anything else (no label, several labels, a variant, a gap, an ambiguous base) takes the loop.
  bool seeded = false;
  if (try_fast && use_halves && n_k > 0 && n_k == kc)
  {
    typename W::template PerLane<bool> bad_l, var_l;
    typename W::template PerLane<uint32_t> mm_l, why_l;
    W::lanes([&](uint32_t l) {
      bool bad = false, has_var = false;
      uint32_t mm = 0, why = 0;
      if (l < n_k)
      {
        uint32_t const c0 = ws.cnt0[l];
        if (ws.nkeys0[l] != 1)
        {
          // One ambiguous base and one label among its keys.  The loop below adds a multi-key list twice (0 and 1
          // mismatches, kmer_help_functions.cpp:97-119); the copy with the mismatch never merges into the chain, is
          // shorter than it (or, from k-mer 0, the same chain with one more mismatch) and is dropped by
          // remove_short_paths / remove_paths_with_too_many_mismatches -- the chain itself is what remains.
          bad = c0 != 1;
          why = 1; // ambiguous base(s) without exactly one label
          if (!bad)
          {
            DevLabel const lb = ws.xl[l][0];
            bad = lb.site != INVALID;
            why = 2; // ambiguous base on a variant
            ws.fs_start[l] = lb.start;
            ws.fs_end[l] = lb.end;
          }
        }
        else if (!(bad = c0 > 1 || ws.hcnt[l][0] > AlignCfg::HE_CAP || ws.hcnt[l][1] > AlignCfg::HE_CAP))
        {
          why = 3;
          uint64_t const q = ws.key0[l];
          uint32_t nb = 0, nb_off = 0;
          for (uint32_t side = 0; side < 2; ++side)
            for (uint32_t e = 0; e < ws.hcnt[l][side]; ++e)
            {
              HalfEntry const & he = ws.he[l][side][e];
              uint32_t j;
              if (hamming1_neighbour(he.key, q, j)) // (a neighbour shares exactly one half with q: it is in one bucket only)
              {
                nb += he.cnt;
                nb_off = he.off;
              }
            }
          if (c0 + nb == 1)
          {
            DevLabel const lb = c0 ? ws.xl[l][0] : ix.labels[nb_off];
            mm = c0 ? 0u : 1u;
            ws.fs_start[l] = lb.start;
            ws.fs_end[l] = lb.end;
            if (lb.site != INVALID) // the only label there is lies on a variant (e.g. an error inside a k-mer over a SNP)
            {
              bad = g.is_sv_graph != 0 || !pv_fits(lb.allele);
              why = 4;
              has_var = !bad;
              ws.fs_site = lb.site;
              ws.fs_allele = lb.allele;
            }
          }
          else if (c0 == 1 && nb <= 3 && !g.is_sv_graph)
          {
            why = 5; // exact label + neighbours that are not "the same SNP"
            // The read carries one allele of a variant site and the only Hamming-1 neighbours are the same interval on
            // the site's other alleles (a SNP).  In the loop below each of those starts a path with one more mismatch
            // that shares every later label and the chain's end with the exact one, so it walks the same tail and
            // loses to it in remove_short_paths / remove_paths_with_too_many_mismatches; the exact chain, carrying
            // {site, allele}, is what remains.  A neighbour anywhere else could out-walk the chain: not handled here.
            DevLabel const lb = ws.xl[l][0];
            bad = lb.site == INVALID || !pv_fits(lb.allele);
            for (uint32_t side = 0; side < 2 && !bad; ++side)
              for (uint32_t e = 0; e < ws.hcnt[l][side]; ++e)
              {
                HalfEntry const & he = ws.he[l][side][e];
                uint32_t j;
                if (hamming1_neighbour(he.key, q, j))
                  for (uint32_t k = 0; k < he.cnt; ++k)
                  {
                    DevLabel const nl = ix.labels[he.off + k];
                    bad = bad || nl.start != lb.start || nl.end != lb.end || nl.site != lb.site;
                  }
              }
            if (!bad)
            {
              has_var = true;
              ws.fs_start[l] = lb.start;
              ws.fs_end[l] = lb.end;
              ws.fs_site = lb.site; // (one writer, or the read is left to the loop)
              ws.fs_allele = lb.allele;
            }
          }
          else
          {
            bad = true;
            why = c0 + nb == 0 ? 6 : 7; // no label at all / several labels
          }
        }
        else
          why = 8; // many exact labels or big half buckets
      }
      bad_l[l] = bad;
      var_l[l] = has_var;
      mm_l[l] = mm;
      why_l[l] = bad ? why : 0u;
    });
#ifdef GTX_EMU_NOTES // diagnostics of the host emulation: why the task leaves the express pass
    {
      uint32_t first_why = 0;
      W::lanes([&](uint32_t l) {
        if (first_why == 0 && why_l[l] != 0)
          first_why = why_l[l];
      });
      if (first_why)
        W::note(first_why);
    }
#endif
    uint64_t const with_var = W::ballot(var_l);
#ifdef GTX_EMU_NOTES
    if (W::ballot(bad_l) == 0 && (with_var & (with_var - 1)) != 0)
      W::note(10); // two k-mers on variants
#endif
    if (W::ballot(bad_l) == 0 && (with_var & (with_var - 1)) == 0)
    {
      W::lds_sync();
      typename W::template PerLane<bool> gap_l;
      W::lanes([&](uint32_t l) { gap_l[l] = l + 1 < n_k && ws.fs_end[l] != ws.fs_start[l + 1]; });
#ifdef GTX_EMU_NOTES
      if (W::ballot(gap_l) != 0)
        W::note(9); // labels do not abut
#endif
      if (W::ballot(gap_l) == 0)
      {
        uint32_t const mism = W::sum(mm_l);
        GTX_LEAD
        {
          DPath & p = ws.paths[0];
          p.start = ws.fs_start[0];
          p.end = ws.fs_end[n_k - 1];
          p.rs = 0;
          p.re = static_cast<uint16_t>((K - 1) * n_k);
          p.mism = static_cast<uint16_t>(mism);
          p.nvar = with_var ? 1 : 0;
          if (with_var)
          {
            p.v[0].site = ws.fs_site;
            pv_set_single(p.v[0], ws.fs_allele);
          }
        }
        W::lds_sync();
        n_paths = 1;
        longest = (K - 1) * n_k + 1;
        seeded = true;
      }
    }
  }
  GTX_PROF_TICK(1)
#ifdef GTX_PROF
  GTX_LEAD ws.prof_acc[14] += seeded ? 1 : 0;
#endif
  return seeded;
}

// Leaves the paths in ws.paths; returns the status bits (non-zero = a table overflowed, the paths are not valid).
// try_fast = false: the first pass already found that this read is not one of the simple ones.
template <class W>
GTX_DEV uint32_t align_paths(GraphView const & g, IndexView const & ix, AlignWorkspace & ws, uint8_t const * seq4, uint32_t len,
                             bool reverse, uint32_t & n_paths_out, uint32_t & longest_out, bool try_fast = true)
{
  GTX_PROF_BEGIN
  uint32_t n_paths = 0, longest = 0, status = 0;
  uint32_t const n_k = len < K ? 0 : 1 + (len - K) / (K - 1); // kmer_help_functions.cpp:10-17
  uint32_t const kc = n_k < AlignCfg::KC ? n_k : AlignCfg::KC;
  bool const use_halves = ix.half_bucket_cap != 0;
  bool const seeded = seed_stage<W>(Here{}, g, ix, ws, seq4, len, reverse, try_fast, n_paths, longest);
  GTX_PROF_RESET
  // -- stop if every k-mer is extremely common (alignment.cpp:35-49); only single-key lists can reach 512 labels
  bool all_common = n_k > 0;
  if (!seeded)
    for (uint32_t i = 0; i < n_k; ++i)
      if (!(GTX_U(ws.nkeys0[i]) == 1 && GTX_U(ws.cnt0[i]) >= MAX_UNIQUE_KMER_POSITIONS))
        all_common = false;
  if (seeded || (!all_common && n_k > 0))
  {
    for (uint32_t i = seeded ? n_k : 0u; i < n_k && !status; ++i)
    {
      uint32_t const rs = (K - 1) * i, re = rs + (K - 1);
      bool const single = GTX_U(ws.nkeys0[i]) == 1;
      uint32_t n_lbl;
      DevLabel const * exact_labels = ws.lbl;
      if (single)
      {
        // exact list: one key, never cut (ph_index.cpp:84)
        uint32_t const cnt = GTX_U(ws.cnt0[i]), off = GTX_U(ws.off0[i]);
        n_lbl = cnt;
        if (cnt > cap_lbl(ws))
        {
          status |= GTX_ST_LABEL_OVERFLOW;
          break;
        }
        if (i < kc && cnt <= AlignCfg::XL_CAP)
          exact_labels = ws.xl[i]; // staged up front
        else
        {
          for (uint32_t b = 0; b < cnt; b += 64)
            W::lanes([&](uint32_t l) {
              if (b + l < cnt)
                ws.lbl[b + l] = ix.labels[off + b + l];
            });
          W::lds_sync();
        }
      }
      else
      {
        uint32_t nk = 0;
        n_lbl = 0;
        uint32_t const amb = GTX_U(ws.off0[i]);
        if ((amb & (amb - 1u)) == 0u && i < AlignCfg::MAX_KMERS && GTX_U(ws.acnt[i][0]) != 0xFFFFFFFFu)
        {
          // One ambiguous base: seed_stage probed the 2..4 keys of its list with all the other lookups of the read (ws.acnt /
          // ws.aoff, in to_uint64_vec order) -- no second visit to the index, the labels are one fetch away (or staged already).
          uint32_t const c0 = GTX_U(ws.acnt[i][0]), c1 = GTX_U(ws.acnt[i][1]), c2 = GTX_U(ws.acnt[i][2]), c3 = GTX_U(ws.acnt[i][3]);
          uint32_t const total = c0 + c1 + c2 + c3;
          n_lbl = total;
          if (total > ix.max_index_labels)
            n_lbl = 0; // multi_get: a list of several keys with more hits than that yields nothing (ph_index.cpp:84-89)
          else if (total > cap_lbl(ws))
          {
            status |= GTX_ST_LABEL_OVERFLOW;
            break;
          }
          else if (total != 0)
          {
            uint32_t const o0 = GTX_U(ws.aoff[i][0]), o1 = GTX_U(ws.aoff[i][1]), o2 = GTX_U(ws.aoff[i][2]), o3 = GTX_U(ws.aoff[i][3]);
            bool const staged = total == 1 && i < kc && GTX_U(ws.cnt0[i]) == 1; // (seed_stage put the one label into ws.xl[i][0])
            for (uint32_t b = 0; b < total; b += 64)
              W::lanes([&](uint32_t l) {
                uint32_t const e = b + l;
                if (e < total)
                  ws.lbl[e] = staged ? ws.xl[i][0]
                                     : ix.labels[e < c0 ? o0 + e : e < c0 + c1 ? o1 + (e - c0) : e < c0 + c1 + c2 ? o2 + (e - c0 - c1) : o3 + (e - c0 - c1 - c2)];
              });
            W::lds_sync();
          }
        }
        else if ((amb & (amb - 1u)) == 0u)
        {
          // one ambiguous base (nearly always a single N): to_uint64_vec's list is the key with the LAST admissible base
          // at that position followed by the other admissible bases in ascending order (type_conversions.cpp:207-266:
          // the last one replaces the key in place, the others are appended) -- one key per lane, no sequential expansion
          uint32_t const t0 = static_cast<uint32_t>(__builtin_ctz(amb));
          uint32_t const code = GTX_U(static_cast<uint32_t>(ws.rd[rs + t0])) & 15u;
          uint32_t const set = (code == 0u || code == 15u) ? 15u : code;
          uint32_t const last = 31u - static_cast<uint32_t>(__builtin_clz(set));
          nk = static_cast<uint32_t>(__builtin_popcount(set));
          uint64_t const base = GTX_U(ws.key0[i]); // the ambiguous base contributed no bits
          W::lanes([&](uint32_t l) {
            if (l < nk)
            {
              uint32_t b = last;
              if (l > 0)
              {
                uint32_t rest = set & ~(1u << last);
                for (uint32_t k = 1; k < l; ++k)
                  rest &= rest - 1u;
                b = static_cast<uint32_t>(__builtin_ctz(rest));
              }
              ws.u.keybuf[l] = base | (static_cast<uint64_t>(b & 1u) << t0) | (static_cast<uint64_t>(b >> 1) << (32u + t0));
            }
          });
          W::lds_sync();
        }
        else if (__builtin_popcount(amb) == 2)
        {
          // Two ambiguous bases A (first in the read) and B: after A the list is [last(A), the other bases of A ascending];
          // B replaces every entry by its version with last(B) and appends, entry by entry, the versions with B's other
          // bases ascending (the same rule applied to a list of |A| keys) -- at most 16 keys, one per lane.
          uint32_t const ta = static_cast<uint32_t>(__builtin_ctz(amb)), tb = 31u - static_cast<uint32_t>(__builtin_clz(amb));
          uint32_t const ca = GTX_U(static_cast<uint32_t>(ws.rd[rs + ta])) & 15u, cb = GTX_U(static_cast<uint32_t>(ws.rd[rs + tb])) & 15u;
          uint32_t const sa = (ca == 0u || ca == 15u) ? 15u : ca, sb = (cb == 0u || cb == 15u) ? 15u : cb;
          uint32_t const la = 31u - static_cast<uint32_t>(__builtin_clz(sa)), lb = 31u - static_cast<uint32_t>(__builtin_clz(sb));
          uint32_t const ma = static_cast<uint32_t>(__builtin_popcount(sa)), mb = static_cast<uint32_t>(__builtin_popcount(sb));
          nk = ma * mb;
          uint64_t const base = GTX_U(ws.key0[i]);
          auto other = [](uint32_t set, uint32_t last, uint32_t k) // the k-th smallest base of the set without its last one
          {
            uint32_t rest = set & ~(1u << last);
            for (uint32_t j = 0; j < k; ++j)
              rest &= rest - 1u;
            return static_cast<uint32_t>(__builtin_ctz(rest));
          };
          W::lanes([&](uint32_t l) {
            if (l < nk)
            {
              uint32_t const u = l < ma ? l : (l - ma) / (mb - 1u);
              uint32_t const ba = u == 0 ? la : other(sa, la, u - 1u);
              uint32_t const bb = l < ma ? lb : other(sb, lb, (l - ma) % (mb - 1u));
              ws.u.keybuf[l] = base | (static_cast<uint64_t>(ba & 1u) << ta) | (static_cast<uint64_t>(ba >> 1) << (32u + ta)) |
                               (static_cast<uint64_t>(bb & 1u) << tb) | (static_cast<uint64_t>(bb >> 1) << (32u + tb));
            }
          });
          W::lds_sync();
        }
        else
        {
          nk = expand_keys_lanes<W>(ws, rs, amb, GTX_U(ws.key0[i]));
          if (nk == 0xFFFFFFFFu)
          {
            status |= GTX_ST_LABEL_OVERFLOW;
            break;
          }
        }
        if (nk != 0)
          n_lbl = probe_list<W>(ix, ws, false, 0, nk, status);
        if (status)
          break;
      }
      GTX_PROF_TICKA(2)
      add_kmer_labels<W>(ws, exact_labels, n_lbl, rs, re, 0, false, n_paths, longest, status);
      GTX_PROF_TICKA(3)
      if (status)
        break;
      // Hamming-1 list: the 96 neighbours of a unique exact key, else the exact list again
      // (kmer_help_functions.cpp:97-119 keeps multi-key lists as they are, so ws.lbl is already what multi_get returns)
      if (single)
      {
        uint64_t const q = GTX_U(ws.key0[i]);
        if (i < kc && use_halves && n_lbl > 0 && GTX_U(ws.hcnt[i][0]) == 1 && GTX_U(ws.hcnt[i][1]) == 1)
          n_lbl = 0; // q is indexed, so it is the one key in each of its half-key buckets: no neighbour exists
        else if (i < kc && use_halves && GTX_U(ws.hcnt[i][0]) <= AlignCfg::HE_CAP && GTX_U(ws.hcnt[i][1]) <= AlignCfg::HE_CAP)
          n_lbl = hamming1_from_cache<W>(ix, ws, i, q);
        else if (!use_halves || !hamming1_by_halves<W>(ix, ws, q, n_lbl))
          n_lbl = probe_list<W>(ix, ws, true, q, 96, status);
        if (n_lbl == 0xFFFFFFFFu) // (hamming1_finish: more labels than the table holds)
        {
          n_lbl = 0;
          status |= GTX_ST_LABEL_OVERFLOW;
        }
        if (status)
          break;
      }
      GTX_PROF_TICKA(4)
      add_kmer_labels<W>(ws, ws.lbl, n_lbl, rs, re, 1, false, n_paths, longest, status);
      GTX_PROF_TICKA(5)
    }
    if (!status && !(seeded && finish_single_path<W>(Here{}, g, ws, n_paths, longest)))
    {
      n_paths = remove_short_paths<W>(ws, n_paths, longest);
      walk_read<W>(g, ws, true, n_paths, longest, status);
      GTX_PROF_TICK(6)
      if (!status)
        walk_read<W>(g, ws, false, n_paths, longest, status);
      GTX_PROF_TICK(7)
      if (!status)
      {
        longest = longest_of<W>(ws, n_paths);
        n_paths = remove_short_paths<W>(ws, n_paths, longest);
        n_paths = remove_paths_with_too_many_mismatches<W>(ws, n_paths);
        if (g.is_sv_graph)
          n_paths = remove_fully_special_paths<W>(g, ws, n_paths);
        n_paths = remove_non_ref_paths_when_read_matches_ref<W>(g, ws, n_paths);
        longest = longest_of<W>(ws, n_paths);
        n_paths = remove_short_paths<W>(ws, n_paths, longest);
        if (g.is_sv_graph)
          remove_support_from_read_ends<W>(g, ws, n_paths);
      }
    }
  }

  GTX_PROF_TICK(8)
#ifdef GTX_PROF
  GTX_LEAD ws.prof_acc[15] += 1;
#endif
  n_paths_out = n_paths;
  longest_out = longest;
  return status;
}

// words of the result record of the paths in ws.paths (layout: include/gtx.h, gtx_align_batch)
template <class W, class WS>
GTX_DEV uint32_t record_size(Here, WS const & ws, uint32_t n_paths)
{
  uint32_t w = 2;
  for (uint32_t i = 0; i < n_paths; ++i)
    w += 4 + PVAR_WORDS * GTX_U(static_cast<uint32_t>(ws.paths[i].nvar));
  return w;
}

// path part of the record (words 2..); `body` has room for record_size() - 2 words
// Returns GTX_REC_HAS_VARIANTS when a path carries a variant site, else 0 (bit 31 of the record's second word: lets the
// scorer skip the 85 % of the reads that cannot add anything without parsing their paths).
template <class W, class WS>
GTX_DEV uint32_t write_record_body(Here, WS const & ws, uint32_t n_paths, uint32_t * body)
{
  uint32_t w = 0, any_var = 0;
  for (uint32_t i = 0; i < n_paths; ++i)
  {
    DPath const & p = ws.paths[i];
    uint32_t const nvar = GTX_U(static_cast<uint32_t>(p.nvar));
    any_var |= nvar;
    uint32_t const * src = reinterpret_cast<uint32_t const *>(&p);
    uint32_t const nw = 4 + PVAR_WORDS * nvar;
    W::lanes([&](uint32_t l) {
      for (uint32_t x = l; x < nw; x += 64)
      {
        uint32_t v = src[x];
        if (x == 3)
          v = static_cast<uint32_t>(p.mism) | (nvar << 16);
        body[w + x] = v;
      }
    });
    w += nw;
  }
  return (any_var ? GTX_REC_HAS_VARIANTS : 0u) | (AlignCfg::MW > 2 ? GTX_REC_WIDE : 0u);
}

// one (read, orientation) of the main pass: result into its record slot
template <class W>
GTX_DEV uint32_t align_one(GraphView const & g, IndexView const & ix, AlignWorkspace & ws, uint8_t const * seq4, uint32_t len,
                           bool reverse, uint32_t * rec, uint32_t rec_words, bool try_fast = true)
{
  uint32_t np = 0, longest = 0;
  uint32_t status = align_paths<W>(g, ix, ws, seq4, len, reverse, np, longest, try_fast);
  if (status)
    np = 0;
  else if (record_size<W>(Here{}, ws, np) > rec_words)
  {
    status = GTX_ST_RECORD_OVERFLOW;
    np = 0;
  }
  GTX_PROF_BEGIN
  uint32_t const has_var = write_record_body<W>(Here{}, ws, np, rec + 2);
  GTX_LEAD
  {
    rec[0] = np | ((status & ~GTX_ST_WIDE_ALLELE) << 16);
    rec[1] = (np == 0 ? 0 : longest) | (len << 16) | has_var;
  }
  W::lds_sync();
  GTX_PROF_TICK(9)
  return status;
}

// First pass, one (read, orientation): the read either is one of the simple ones -- fast seeding gives its single path and
// finish_single_path completes it -- and its record is written, or nothing is written and the caller queues it for the
// general pass (returns false).
template <class W>
GTX_DEV bool express_one(GraphView const & g, IndexView const & ix, SeedWorkspace & ws, uint8_t const * seq4, uint32_t len,
                         bool reverse, uint32_t * rec, uint32_t rec_words)
{
  uint32_t np = 0, longest = 0;
  if (!seed_stage<W>(Here{}, g, ix, ws, seq4, len, reverse, true, np, longest))
    return false;
  if (!finish_single_path<W>(Here{}, g, ws, np, longest))
  {
#ifdef GTX_EMU_NOTES
    W::note(11); // the tail does not fit in the reference node the chain ends in
#endif
    return false;
  }
  if (record_size<W>(Here{}, ws, np) > rec_words)
    return false;
  GTX_PROF_BEGIN
  uint32_t const has_var = write_record_body<W>(Here{}, ws, np, rec + 2);
  GTX_LEAD
  {
    rec[0] = np;
    rec[1] = (np == 0 ? 0 : longest) | (len << 16) | has_var;
  }
  W::lds_sync();
  GTX_PROF_TICK(9)
#ifdef GTX_PROF
  GTX_LEAD ws.prof_acc[15] += 1;
#endif
  return true;
}
