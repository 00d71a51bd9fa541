// express4.inl -- pass 1 of the alignment with FOUR reads per wavefront (16 lanes each).
//
// Same decisions as express_one (align_core.inl: seed_stage with try_fast + finish_single_path), restated so that one
// instruction stream serves four independent reads: a simple read needs few lanes (5 k-mers, 15 lookups, a 26-base tail),
// so a wave of 64 spent most of its issue slots on one read's scalar bookkeeping and overlapped nothing; here four
// latency chains run side by side and the bookkeeping is shared.  Included from align_core.hpp inside namespace gtx
// (main-pass table sizes).  Group gi = lane >> 4 works on task `first + gi` (forward orientation of one read).
//
// A group either writes the read's record (same words express_one writes) or reports that the task needs pass 2;
// it never writes a partial result.  What it declines: reads over 187 bp (more than KC k-mers), k-mers with several
// ambiguous bases, every case seed_stage's fast seeding declines (except a single k-mer without any label, which it
// handles: see the run selection below), walks that leave the reference node.

// Two builds of the pass.  The lean one is for graphs whose sites lie far apart (a k-mer meets at most one site, the walk
// at the read's end at most one); the wide one, for dense graphs, takes k-mers with several labels on one interval (a
// k-mer over up to KS sites) and tails over up to TS SNP-like sites, at the price of more registers and LDS.  They
// decide the same reads the same way where both accept; the wide one merely declines fewer.
struct Express4Lean
{
  static constexpr uint32_t KS = 1;                 // variant sites of one k-mer
  static constexpr uint32_t NB_MAX = 3;             // labels of a k-mer's Hamming-1 neighbours that are looked at
  static constexpr uint32_t TS = 1;                 // variant sites under the walk at the read's end
  static constexpr uint32_t VS_CAP = AlignCfg::KC + 1; // variant sites of the path
  static constexpr bool END_ON_SITE = false;        // paths whose last base lies on a SNP are walked on here
  static constexpr bool INDEL_TAIL = false;         // walks at the read's end over a site with alleles of any length
  static constexpr uint32_t HE_SCAN = 0;            // entries of a crowded half-key bucket that are searched for the neighbours
  static constexpr uint32_t AMB_LABELS = 1;         // labels a k-mer with an ambiguous base may have between its keys
  static constexpr bool AMB_ON_VARIANT = true;      // ... and whether they may lie on a variant
};

static_assert(Express4Lean::KS == 1, "one variant site per k-mer in the position flags");

struct Express4Wide
{
  static constexpr uint32_t KS = 4, NB_MAX = 16, TS = 3, VS_CAP = 16;
  static constexpr bool END_ON_SITE = true;
  static constexpr bool INDEL_TAIL = true;
  static constexpr uint32_t HE_SCAN = 16;
  static constexpr uint32_t AMB_LABELS = 5;
  static constexpr bool AMB_ON_VARIANT = true;
};

static_assert(Express4Wide::HE_SCAN == HINT_HE_CAP && Express4Wide::NB_MAX == HINT_NB_MAX,
              "hint_exact_verdict (index_build.hpp) restates the wide seeding rule with these limits");

constexpr uint32_t EXPRESS4_INDEL_ALLELES = 8; // alleles of a site the walk at the read's end may cross when they differ in length

template <bool ON>
struct Express4IndelTail // in the tail handed to the lanes
{
  uint32_t indel, cmp_len;
};

template <>
struct Express4IndelTail<false> // (the lean build carries none of it)
{
};

template <bool ON>
struct Express4IndelAlleles // written by the leader lane only when it meets such a site
{
  uint32_t next_off, next_order, site_order; // (the alleles of a site share their order)
  uint32_t alen[EXPRESS4_INDEL_ALLELES], aoff[EXPRESS4_INDEL_ALLELES];
  uint32_t amm[EXPRESS4_INDEL_ALLELES]; // mismatches of allele a's candidate behind the common part (summed by the leader lane)
};

template <>
struct Express4IndelAlleles<false>
{
};

template <class E4>
struct Express4Tail // handed from a group's leader lane to its 16 lanes
{
  uint32_t tail_len, pre;           // walk at the read's end: characters (from the path's last base on), read offset
  uint32_t head_off, head_len, prs; // walk at the read's start (backwards from the path's first base)
  uint32_t ok;
  // A tail may run over SNP-like sites (every allele one base).  Tail character site_at[k] lies on site k; the characters
  // before site 0 lie in the reference node the path ends in, those after site k in the reference node that follows it:
  // character i of stretch k sits at arena offset seg_base[k] + i.
  uint32_t nsite;
  uint32_t site_at[E4::TS], seg_base[E4::TS + 1];
  uint32_t nall[E4::TS], alleles[E4::TS], site[E4::TS]; // alleles: one comparison code per byte
  uint32_t afirst[E4::TS];                              // index of the first of them (0 unless the walk starts inside an allele)
  // A tail over ONE site with alleles of any length (an indel): characters [0, x.cmp_len) lie in the node the path ends
  // in, then allele a's bases, then the reference node after the site (Express4IndelAlleles in the workspace).  One
  // candidate per allele, compared separately; site[0] / nall[0] name the site.
  Express4IndelTail<E4::INDEL_TAIL> x;
};

template <class E4>
struct Express4Workspace
{
  SeedWorkspace s[4];
  Express4Tail<E4> tail[4];
  Express4IndelAlleles<E4::INDEL_TAIL> xtail[4];
  // the variant sites of every k-mer's labels, in label order, and the alleles the labels name
  uint32_t kn[4][AlignCfg::KC], ksite[4][AlignCfg::KC][E4::KS];
  uint64_t kmask[4][AlignCfg::KC][E4::KS];
  uint32_t vsite[4][E4::VS_CAP]; // the path's sites in record order (built by the leader lane)
  uint64_t vmask[4][E4::VS_CAP];
  uint8_t nbk[4][8];             // SLOT_NB_KNOWN of every k-mer's exact slot (the neighbours' labels need not be fetched)
#ifdef GTX_PROF
  unsigned long long prof_acc[16]; // phase cycles of the profiling build (libgtx_prof.so)
#endif
};

#ifdef GTX_EMU_NOTES // diagnostics of the host emulation (tests/emu): why a task leaves this pass
#define GTX_E4_NOTE(cond, code)                                                                                                    \
  do                                                                                                                               \
  {                                                                                                                                \
    if (cond)                                                                                                                      \
      W::note(code);                                                                                                               \
  } while (0)
#else
#define GTX_E4_NOTE(cond, code)                                                                                                    \
  do                                                                                                                               \
  {                                                                                                                                \
  } while (0)
#endif

// Returns a 4-bit mask: bit gi set = task first + gi must go through pass 2.  With `rid` the four reads are rid[0..3]
// (a slice of the queue the position-hinted pass filled) instead of first .. first + 3.
template <class W, class E4>
GTX_DEV uint32_t express4(GraphView const & g, IndexView const & ix, Express4Workspace<E4> & ws, uint8_t const * seq, uint32_t seq_stride,
                          gtx_read_meta const * meta, uint32_t first, uint32_t n_valid, uint32_t * records, uint32_t rec_words,
                          bool decline_all = false, uint32_t const * rid = nullptr)
{
  using PB = typename W::template PerLane<bool>;
  using PU = typename W::template PerLane<uint32_t>;
  constexpr uint32_t KC = AlignCfg::KC, HE_CAP = AlignCfg::HE_CAP;
  bool const use_halves = ix.half_bucket_cap != 0;
  GTX_PROF_BEGIN

  // ---- per group: the read, whether it is a candidate at all
  PB alive_l, pass2_l;
  PU len_l, nk_l, read_l;
  // (the lane's share of the read's bases -- four bases per round of the unpacking below -- is fetched together with the
  // read's length: both hang on the read number only, and fetching the bases behind the length was a third of this pass'
  // first phase)
  static_assert(AlignCfg::MAX_READ / 64 == 4, "four rounds of 64 bases: two 32-bit words of plane nibbles per lane");
  PU raw01_l, raw23_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4;
    bool const valid = gi < n_valid;
    uint32_t const read = !valid ? 0u : rid ? rid[gi] : first + gi;
    read_l[l] = read;
    uint32_t const len = valid ? static_cast<uint32_t>(meta[read].l_qseq) : 0u;
    {
      // the read is a row of bit planes (graph_dev.hpp): the lane's four bases of round `it` are four bits of each of the
      // four plane words of one 16-byte group; they are kept as four nibbles (plane 0 lowest) per round
      uint4_t const * row = reinterpret_cast<uint4_t const *>(seq + static_cast<uint64_t>(read) * seq_stride);
      uint32_t raw[4];
      for (uint32_t it = 0; it < 4; ++it)
      {
        uint32_t const base = 4 * (it * 16 + (l & 15u)), grp = base >> 5, o = base & 31u;
        uint4_t gq{0, 0, 0, 0};
        if (valid && (grp + 1) * PLANE_GROUP_BYTES <= seq_stride)
          gq = row[grp];
        raw[it] = ((gq.x >> o) & 15u) | (((gq.y >> o) & 15u) << 4) | (((gq.z >> o) & 15u) << 8) | (((gq.w >> o) & 15u) << 12);
      }
      raw01_l[l] = raw[0] | (raw[1] << 16);
      raw23_l[l] = raw[2] | (raw[3] << 16);
    }
    bool const too_short = len < 2 * K - 1, too_long = len > AlignCfg::MAX_READ;
    uint32_t const n_k = len < K ? 0 : 1 + (len - K) / (K - 1);
    if (valid && (too_short || too_long) && (l & 15u) == 0)
    {
      // align_read (alignment.cpp:331-363): reads shorter than 2K-1 stay unaligned
      uint32_t * rec = records + static_cast<uint64_t>(read) * 2 * rec_words;
      rec[0] = too_long ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
      rec[1] = len << 16;
    }
    bool const candidate = valid && !too_short && !too_long;
    bool const fits = candidate && n_k <= KC && use_halves && rec_words >= 9 && !decline_all; // (decline_all: test switch)
    GTX_E4_NOTE((l & 15u) == 0 && candidate && !fits, 11);
    alive_l[l] = fits;
    pass2_l[l] = candidate && !fits;
    len_l[l] = len;
    nk_l[l] = n_k;
  });

  if (decline_all)
  {
    uint64_t const P2 = W::ballot(pass2_l);
    return (P2 & 1ull ? 1u : 0u) | (P2 >> 16 & 1ull ? 2u : 0u) | (P2 >> 32 & 1ull ? 4u : 0u) | (P2 >> 48 & 1ull ? 8u : 0u);
  }

  // ---- unpack the reads (bit planes -> one code per byte), 4 bases per lane and round
  for (uint32_t it = 0; it < AlignCfg::MAX_READ / 64; ++it)
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4, word = it * 16 + (l & 15u), len = len_l[l];
      if (alive_l[l] && 4 * word < len)
      {
        uint32_t const four = ((it < 2 ? raw01_l[l] : raw23_l[l]) >> (16 * (it & 1u))) & 0xFFFFu; // four bases as four plane nibbles
        uint32_t packed = plane_spread4(four & 15u) | (plane_spread4((four >> 4) & 15u) << 1) | (plane_spread4((four >> 8) & 15u) << 2) |
                          (plane_spread4((four >> 12) & 15u) << 3);
        // bases behind the read's end count as N, and so does '=' (assigned to a seqan Iupac it becomes N,
        // hts_parallel_reader.cpp:226-243): both are zero bytes by now
        uint32_t const inside = len - 4 * word >= 4 ? 0xFFFFFFFFu : (1u << (8 * (len - 4 * word))) - 1u;
        packed &= inside;
        uint32_t const zero = ~(packed | (packed >> 1) | (packed >> 2) | (packed >> 3)) & 0x01010101u;
        packed |= zero * 15u;
        reinterpret_cast<uint32_t *>(ws.s[gi].rd)[word] = packed;
      }
    });
  W::lds_sync();
  GTX_PROF_TICK(0)

  // ---- exact keys in plane form: two rounds of 16 bases per k-mer, each ballot serves the four groups
  for (uint32_t i = 0; i < KC; ++i)
  {
    PU lo_l, hi_l, amb_l;
    for (uint32_t h = 0; h < 2; ++h)
    {
      PB a_l, b0_l, b1_l;
      W::lanes([&](uint32_t l) {
        uint32_t const gi = l >> 4, j = l & 15u;
        uint32_t const c = (alive_l[l] && i < nk_l[l]) ? ws.s[gi].rd[(K - 1) * i + 16 * h + j] : 1u;
        bool const single = (c & (c - 1u)) == 0u && c != 0u;
        uint32_t const two = (c == 2u) ? 1u : (c == 4u) ? 2u : (c == 8u) ? 3u : 0u;
        a_l[l] = !single;
        b0_l[l] = (two & 1u) != 0u;
        b1_l[l] = (two & 2u) != 0u;
      });
      uint64_t const A = W::ballot(a_l), B0 = W::ballot(b0_l), B1 = W::ballot(b1_l);
      W::lanes([&](uint32_t l) {
        uint32_t const sh = 16 * (l >> 4);
        uint32_t const a = static_cast<uint32_t>(A >> sh) & 0xFFFFu, b0 = static_cast<uint32_t>(B0 >> sh) & 0xFFFFu,
                       b1 = static_cast<uint32_t>(B1 >> sh) & 0xFFFFu;
        lo_l[l] = (h ? lo_l[l] : 0u) | (b0 << (16 * h));
        hi_l[l] = (h ? hi_l[l] : 0u) | (b1 << (16 * h));
        amb_l[l] = (h ? amb_l[l] : 0u) | (a << (16 * h));
      });
    }
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4;
      if ((l & 15u) == 0 && alive_l[l] && i < nk_l[l])
      {
        SeedWorkspace & s = ws.s[gi];
        s.key0[i] = (static_cast<uint64_t>(hi_l[l]) << 32) | lo_l[l];
        s.nkeys0[i] = amb_l[l] == 0 ? 1 : 2;
        s.cnt0[i] = 0;
        s.off0[i] = amb_l[l];
        s.hcnt[i][0] = 0;
        s.hcnt[i][1] = 0;
      }
    });
  }
  W::lds_sync();
  GTX_PROF_TICK(1)

  // ---- index lookups: lane j of a group = (k-mer j / 3, exact | left half | right half); a single label / bucket entry
  //      comes inline with its slot
  PB amb_any_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4, j = l & 15u, i = j / 3, w = j % 3;
    SeedWorkspace & s = ws.s[gi];
    bool amb_here = false;
    if (alive_l[l] && j < 3 * nk_l[l])
    {
      if (s.nkeys0[i] == 1)
      {
        uint64_t const q = s.key0[i];
        uint32_t off, cnt, slot_flags;
        IndexSlot const * hit;
        bucket_find(w == 0 ? ix.slots : ix.hslots, w == 0 ? ix.log2_cap : ix.h_log2_cap, w == 0 ? q : half_key(q, w - 1), off, cnt, &hit, &slot_flags);
        if (w == 0)
        {
          s.off0[i] = off;
          s.cnt0[i] = cnt;
          ws.nbk[gi][i] = (slot_flags & SLOT_NB_KNOWN) ? 1 : 0;
        }
        else
        {
          s.hoff[i][w - 1] = off;
          s.hcnt[i][w - 1] = cnt;
        }
        if (cnt == 1)
        {
          uint4_t const payload = *reinterpret_cast<uint4_t const *>(hit->p);
          *(w == 0 ? reinterpret_cast<uint4_t *>(&s.xl[i][0]) : reinterpret_cast<uint4_t *>(&s.he[i][w - 1][0])) = payload;
        }
      }
      else
        amb_here = w == 0;
    }
    amb_any_l[l] = amb_here;
  });
  if (W::ballot(amb_any_l) != 0)
  {
    // k-mers with one ambiguous base: lane j = (k-mer j >> 2, key j & 3) in to_uint64_vec order; the one label the keys
    // may have between them is taken from its slot.  (k-mer 4 of a 156..187 bp read has no lane here: left to pass 2.)
    W::lds_sync();
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4, j = l & 15u, ai = j >> 2, aw = j & 3u;
      SeedWorkspace & s = ws.s[gi];
      if (alive_l[l] && ai < nk_l[l] && s.nkeys0[ai] != 1)
      {
        uint32_t const amb = s.off0[ai];
        uint32_t off = 0, cnt = aw == 0 ? 0xFFFFFFFFu : 0u; // several ambiguous bases: "unknown"
        if ((amb & (amb - 1u)) == 0u)
        {
          uint32_t const t0 = static_cast<uint32_t>(__builtin_ctz(amb));
          uint32_t const code = s.rd[(K - 1) * ai + t0] & 15u;
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
            uint64_t const key = s.key0[ai] | (static_cast<uint64_t>(b & 1u) << t0) | (static_cast<uint64_t>(b >> 1) << (32u + t0));
            IndexSlot const * hit;
            bucket_find(ix.slots, ix.log2_cap, key, off, cnt, &hit);
            if (cnt == 1)
              s.xl[ai][0] = DevLabel{hit->p[0], hit->p[1], hit->p[2], hit->p[3]}; // (matters only if it is the only one)
          }
        }
        s.acnt[ai][aw] = cnt;
        s.aoff[ai][aw] = off;
      }
    });
  }
  W::lds_sync();
  GTX_PROF_TICK(2)

  // ---- half-key buckets with 2..HE_CAP entries (a SNP under the k-mer, an error next to one): fetch the entries
  {
    PB need_l;
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4, j = l & 15u;
      bool need = false;
      if (alive_l[l] && j < 2 * nk_l[l])
      {
        uint32_t const c = ws.s[gi].hcnt[j >> 1][j & 1u];
        need = ws.s[gi].nkeys0[j >> 1] == 1 && c >= 2 && c <= HE_CAP;
      }
      need_l[l] = need;
    });
    if (W::ballot(need_l) != 0)
    {
      for (uint32_t e = 0; e < HE_CAP; ++e)
        W::lanes([&](uint32_t l) {
          uint32_t const gi = l >> 4, j = l & 15u;
          if (need_l[l] && e < ws.s[gi].hcnt[j >> 1][j & 1u])
            ws.s[gi].he[j >> 1][j & 1u][e] = ix.hlist[ws.s[gi].hoff[j >> 1][j & 1u] + e];
        });
      W::lds_sync();
    }
    if constexpr (E4::HE_SCAN != 0)
    {
      // Crowded buckets (sites a few bases apart: every combination of their alleles shares the 16-mer beside them): only
      // the Hamming-1 neighbours of the k-mer matter; the bucket is read through and up to HE_CAP of them are kept.
      PB scan_l;
      W::lanes([&](uint32_t l) {
        uint32_t const gi = l >> 4, j = l & 15u;
        bool scan = false;
        if (alive_l[l] && j < 2 * nk_l[l])
        {
          uint32_t const c = ws.s[gi].hcnt[j >> 1][j & 1u];
          scan = ws.s[gi].nkeys0[j >> 1] == 1 && c > HE_CAP && c <= E4::HE_SCAN;
        }
        scan_l[l] = scan;
      });
      if (W::ballot(scan_l) != 0)
      {
        W::lanes([&](uint32_t l) {
          uint32_t const gi = l >> 4, j = l & 15u;
          if (scan_l[l])
          {
            SeedWorkspace & s = ws.s[gi];
            uint64_t const q = s.key0[j >> 1];
            uint32_t const c = s.hcnt[j >> 1][j & 1u], off = s.hoff[j >> 1][j & 1u];
            uint32_t kept = 0;
            for (uint32_t e = 0; e < E4::HE_SCAN; e += 2)
            {
              // (two at a time: independent loads)
              HalfEntry const a = ix.hlist[off + (e < c ? e : 0u)], b = ix.hlist[off + (e + 1 < c ? e + 1 : 0u)];
              uint32_t nn;
              if (e < c && hamming1_neighbour(a.key, q, nn))
              {
                if (kept < HE_CAP)
                  s.he[j >> 1][j & 1u][kept] = a;
                ++kept;
              }
              if (e + 1 < c && hamming1_neighbour(b.key, q, nn))
              {
                if (kept < HE_CAP)
                  s.he[j >> 1][j & 1u][kept] = b;
                ++kept;
              }
            }
            s.hcnt[j >> 1][j & 1u] = kept; // (more than HE_CAP: the k-mer is declined below)
          }
        });
        W::lds_sync();
      }
    }
  }

  GTX_PROF_TICK(3)
  // ---- fast seeding, lane j < n_k of a group = k-mer j (the rules and their justification: seed_stage).
  //      One k-mer of the read may have no label at all (two or more errors, an error next to an N): a "hole".
  PB bad_l, var_l, mm_l, hole_l, par_l; // par: the k-mer also starts a parallel (+1 mismatch) chain when it opens a run
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4, j = l & 15u;
    SeedWorkspace & s = ws.s[gi];
    bool bad = false, has_var = false, mm = false, hole = false, par = false;
    if (alive_l[l] && j < nk_l[l])
    {
      // The labels that make the k-mer's path have to share (start, end): labels with equal ends are one path
      // (find_all_nonduplicated_paths, genotype_paths.cpp:32-66) whose allele set per site is the union over the labels
      // (Path::merge_with_current, path.cpp:105-129); sites keep the order of their first label.
      // (every loop below has a compile-time trip count: the lean build's collapse to straight code)
      uint32_t nsites = 0;
      uint32_t ks[E4::KS];
      uint64_t km[E4::KS];
      for (uint32_t i = 0; i < E4::KS; ++i)
      {
        ks[i] = INVALID;
        km[i] = 0;
      }
      uint32_t l0_start = 0, l0_end = 0;
      bool first_label = true;
      auto take = [&](DevLabel const & lb, bool only_one)
      {
        if (first_label)
        {
          l0_start = lb.start;
          l0_end = lb.end;
          first_label = false;
        }
        bad = bad || lb.start != l0_start || lb.end != l0_end || (lb.site == INVALID && !only_one) ||
              (lb.site != INVALID && lb.allele >= 64u); // (an allele beyond the 64-bit sets of this pass: left to the general passes)
        if (!bad && lb.site != INVALID)
        {
          bool placed = false;
          for (uint32_t i = 0; i < E4::KS; ++i)
            if (!placed && (ks[i] == lb.site || i == nsites))
            {
              ks[i] = lb.site;
              km[i] |= 1ull << lb.allele;
              placed = true;
              nsites = i == nsites ? nsites + 1 : nsites;
            }
          bad = !placed; // more sites than the build takes
        }
      };
      uint32_t const c0 = s.cnt0[j];
      if (s.nkeys0[j] != 1)
      {
        // one ambiguous base: the labels of its (up to four) keys in to_uint64_vec order.  A multi-key list gets no
        // Hamming-1 lookup and is added twice (0 and 1 mismatches): a parallel chain.
        uint32_t const a0 = j < 4 ? s.acnt[j][0] : 0xFFFFFFFFu;
        uint32_t total = a0;
        for (uint32_t aw = 1; aw < 4; ++aw)
          total = (total == 0xFFFFFFFFu || j >= 4) ? 0xFFFFFFFFu : total + s.acnt[j][aw];
        hole = total == 0;
        bad = !hole && total > E4::AMB_LABELS;
        par = true;
        if (!bad && !hole)
        {
          if (total == 1)
            take(s.xl[j][0], true);
          else if (E4::AMB_LABELS > 1)
            for (uint32_t aw = 0; aw < 4; ++aw)
              for (uint32_t k = 0; k < s.acnt[j][aw] && !bad; ++k)
                take(ix.labels[s.aoff[j][aw] + k], false);
        }
      }
      else if (!(bad = c0 > E4::KS || s.hcnt[j][0] > HE_CAP || s.hcnt[j][1] > HE_CAP))
      {
        uint64_t const q = s.key0[j];
        uint32_t nb = 0, nb_keys = 0, nb_off = 0;
        for (uint32_t side = 0; side < 2; ++side)
          for (uint32_t e = 0; e < s.hcnt[j][side]; ++e)
          {
            HalfEntry const & he = s.he[j][side][e];
            uint32_t nn;
            if (hamming1_neighbour(he.key, q, nn))
            {
              nb += he.cnt;
              ++nb_keys;
              nb_off = he.off;
            }
          }
        // the labels of the exact key, else -- no exact hit, ONE neighbouring key -- that key's (one more mismatch)
        uint32_t const n_own = c0 ? c0 : nb;
        if (c0 + nb == 0)
          hole = true;
        else if ((c0 == 0 && nb_keys != 1) || n_own > E4::KS || nb > E4::NB_MAX)
        {
          bad = true;
          GTX_E4_NOTE(true, 2); // neighbours of several keys without an exact hit / too many labels
        }
        else
        {
          mm = c0 == 0;
          uint32_t const own_off = c0 ? s.off0[j] : nb_off;
          for (uint32_t k = 0; k < E4::KS; ++k)
            if (k < n_own && !bad)
              take(c0 == 1 ? s.xl[j][0] : ix.labels[own_off + k], n_own == 1);
          if (!bad && c0 != 0 && nb != 0)
          {
            // Neighbouring keys of an exact hit: the sites' other alleles.  They start chains with one more mismatch
            // that end where the exact chain ends, as long as they are the same interval over the same sites.
            bad = nsites == 0;
            // (SLOT_NB_KNOWN: the index build has looked at the neighbours' labels already -- they pass)
            for (uint32_t side = 0; side < 2 && !bad && !ws.nbk[gi][j]; ++side)
              for (uint32_t e = 0; e < s.hcnt[j][side]; ++e)
              {
                HalfEntry const & he = s.he[j][side][e];
                uint32_t nn;
                if (hamming1_neighbour(he.key, q, nn))
                  for (uint32_t k = 0; k < he.cnt; ++k)
                  {
                    DevLabel const nl = ix.labels[he.off + k];
                    bool known = false;
                    for (uint32_t i = 0; i < E4::KS; ++i)
                      known = known || (i < nsites && ks[i] == nl.site);
                    bad = bad || nl.start != l0_start || nl.end != l0_end || !known;
                  }
              }
            par = true;
          }
          GTX_E4_NOTE(bad, 3); // the k-mer's labels are not one interval / its neighbours lie elsewhere
        }
      }
      // labels on variants (also the only ones there are: e.g. an error inside a k-mer over a SNP)
      bad = bad || (nsites != 0 && (g.is_sv_graph != 0 || (s.nkeys0[j] != 1 && !E4::AMB_ON_VARIANT)));
      has_var = !bad && !hole && nsites != 0;
      if (!bad && !hole)
      {
        s.fs_start[j] = l0_start;
        s.fs_end[j] = l0_end;
      }
      if (has_var)
      {
        ws.kn[gi][j] = nsites;
        for (uint32_t i = 0; i < E4::KS; ++i)
          if (i < nsites)
          {
            ws.ksite[gi][j][i] = ks[i];
            ws.kmask[gi][j][i] = km[i];
          }
      }
      GTX_E4_NOTE(bad && s.nkeys0[j] != 1, 1); // a k-mer with ambiguous bases: several intervals / too many labels
      GTX_E4_NOTE(bad && s.nkeys0[j] == 1 && c0 > E4::KS, 4); // more exact labels than the build takes
      GTX_E4_NOTE(bad && s.nkeys0[j] == 1 && c0 <= E4::KS && (s.hcnt[j][0] > HE_CAP || s.hcnt[j][1] > HE_CAP), 5); // a crowded half-key bucket
    }
    bad_l[l] = bad;
    var_l[l] = has_var;
    mm_l[l] = mm;
    hole_l[l] = hole;
    par_l[l] = par;
  });
  uint64_t const BAD = W::ballot(bad_l), VAR = W::ballot(var_l), MM = W::ballot(mm_l), HOLE = W::ballot(hole_l), PAR = W::ballot(par_l);
  W::lds_sync();
  GTX_PROF_TICK(4)
  // ---- the run of k-mers that makes the path: all of them, or -- with one hole -- the longer side of the hole.  (The
  //      shorter side chains into a shorter path that remove_short_paths drops before the walks, genotype_paths.cpp:
  //      824-834; equal sides would both survive: left to pass 2.)  Any number of the run's k-mers may lie on a variant
  //      (one site per k-mer): the chain collects the sites, most recent first (path.cpp:38-82 keeps p2's sites first).
  PU lo_l, hi_l; // first / last k-mer of the run
  PB run_ok_l;
  W::lanes([&](uint32_t l) {
    uint32_t const sh = 16 * (l >> 4), n_k = nk_l[l];
    uint32_t const bad = static_cast<uint32_t>(BAD >> sh) & 0xFFFFu, var = static_cast<uint32_t>(VAR >> sh) & 0xFFFFu,
                   hole = static_cast<uint32_t>(HOLE >> sh) & 0xFFFFu;
    (void)var;
    bool ok = alive_l[l] && bad == 0;
    uint32_t lo = 0, hi = n_k ? n_k - 1 : 0;
    if (ok && hole != 0)
    {
      // the longest run of k-mers with labels; it has to be the only one of its length
      uint32_t best_lo = 0, best_len = 0, second = 0, cur_lo = 0, cur_len = 0;
      for (uint32_t k = 0; k <= KC; ++k)
      {
        bool const labelled = k < n_k && ((hole >> k) & 1u) == 0;
        if (labelled)
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
      ok = best_len > second;
      lo = best_lo;
      hi = best_lo + (best_len ? best_len - 1 : 0);
      // A run that starts (after a hole) with a multi-key k-mer, or with a k-mer on a variant whose other alleles are
      // indexed, starts with parallel chains (the list's +1-mismatch copy, the other alleles); the start walk then yields
      // one label list per chain, the later lists find no chain left to merge with, become paths of their own and are
      // walked to duplicates of the result (the reference really returns the path twice): left to pass 2.
      if (ok && lo > 0 && ((static_cast<uint32_t>(PAR >> sh) >> lo) & 1u))
        ok = false;
    }
    GTX_E4_NOTE((l & 15u) == 0 && alive_l[l] && bad == 0 && !ok, 6); // two holes, equal sides, parallel chains after the hole
    run_ok_l[l] = ok;
    lo_l[l] = lo;
    hi_l[l] = hi;
  });
  PB gap_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4, j = l & 15u;
    gap_l[l] = run_ok_l[l] && j >= lo_l[l] && j < hi_l[l] && ws.s[gi].fs_end[j] != ws.s[gi].fs_start[j + 1];
  });
  uint64_t const GAP = W::ballot(gap_l);

  // ---- the single path and the geometry of the walks at its two ends (leader lane of every group).  Both walks are
  //      the shortcut of walk_read: the missing part of the read has to lie in the reference node the path touches.
  PB seeded_l;
  PU mism_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4, sh = 16 * gi;
    uint32_t const gap = static_cast<uint32_t>(GAP >> sh) & 0xFFFFu, mm = static_cast<uint32_t>(MM >> sh) & 0xFFFFu;
    bool const seeded = run_ok_l[l] && gap == 0;
    GTX_E4_NOTE((l & 15u) == 0 && run_ok_l[l] && gap != 0, 7); // labels do not abut
    seeded_l[l] = seeded;
    uint32_t const lo = lo_l[l], hi = hi_l[l];
    uint32_t const run_mask = seeded ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    mism_l[l] = static_cast<uint32_t>(__builtin_popcount(mm & run_mask));
    if ((l & 15u) == 0)
    {
      Express4Tail<E4> t{};
      for (uint32_t k = 0; k < E4::TS; ++k)
        t.site_at[k] = INVALID;
      if (seeded)
      {
        uint32_t const L = len_l[l], prs = (K - 1) * lo, pre = (K - 1) * (hi + 1);
        t.pre = pre;
        t.prs = prs;
        t.ok = 1;
        bool const table = g.pos_info && g.n_ref > 1;
        if (prs != 0) // walk_read_starts: read bases 0..prs against the node, backwards from the path's start
        {
          uint32_t const anchor = ws.s[gi].fs_start[lo];
          t.ok = 0;
          if (table && !g_is_special(g, anchor) && anchor >= g.first_order && anchor - g.first_order < g.n_pos_info)
          {
            uint32_t const w = g.pos_info[anchor - g.first_order];
            if (w != INVALID && static_cast<uint32_t>(g.pos_back[anchor - g.first_order]) >= prs && prs + 1 <= 255)
            {
              t.ok = 1;
              t.head_off = w >> 8;
              t.head_len = prs + 1;
            }
          }
        }
        if (t.ok && pre != L - 1) // walk_read_ends (finish_single_path)
        {
          uint32_t const anchor = ws.s[gi].fs_end[hi];
          t.ok = 0;
          if (table && !g_is_special(g, anchor) && anchor >= g.first_order && anchor - g.first_order < g.n_pos_info)
          {
            uint32_t const w = g.pos_info[anchor - g.first_order];
            uint32_t const tail_len = L - pre;
            t.seg_base[0] = w >> 8;
            if (w != INVALID && (w & 255u) >= tail_len)
            {
              t.ok = 1;
              t.tail_len = tail_len;
              if constexpr (E4::INDEL_TAIL)
                t.x.cmp_len = tail_len;
            }
            else if (((E4::END_ON_SITE && w == INVALID) || (w != INVALID && (w & 255u) < 255u)) && !g.is_sv_graph && g.pos_node)
            {
              // The tail leaves the node over variant sites.  When every allele of such a site is a single base (a SNP)
              // and the rest fits in the reference node after the last of them, Graph::get_labels_forward has one
              // candidate per combination of alleles, they differ in those characters only, and the labels of the best
              // ones share (start, end): one path whose allele sets are the best alleles of every site (make_pp unites
              // labels with equal ends).
              uint32_t r = INVALID;
              uint32_t at = 0; // tail character on the next site
              uint32_t n = 0;
              bool done = false;
              if (w != INVALID)
              {
                r = g.pos_node[anchor - g.first_order];
                at = w & 255u;
              }
              else if (E4::END_ON_SITE)
              {
                // The path ends ON a variant base: the walk starts inside the allele the path carries there (Graph::
                // get_locations_of_a_position offers variant nodes the path has, graph.cpp:1154-1185), its first
                // character is that base again, and its labels name the allele: the site moves to the front of the list.
                uint32_t const hi_sites = ((VAR >> (16 * gi + hi)) & 1ull) ? ws.kn[gi][hi] : 0u;
                for (uint32_t i = 0; i < hi_sites && n == 0; ++i)
                {
                  uint32_t const site = ws.ksite[gi][hi][i];
                  uint64_t const mask = ws.kmask[gi][hi][i];
                  uint32_t const nv = g.ref_nvar[site], fv = g.ref_first_var[site];
                  if (g.var_order[fv] != anchor || (mask & (mask - 1ull)) != 0 || site + 1 >= g.n_ref)
                    continue;
                  bool snp = nv <= 4;
                  for (uint32_t a = 0; a < 4 && snp; ++a)
                    if (a < nv)
                      snp = g.var_len[fv + a] == 1;
                  if (!snp)
                    break;
                  uint32_t const a = static_cast<uint32_t>(__builtin_ctzll(mask));
                  t.site_at[0] = 0;
                  t.nall[0] = 1;
                  t.afirst[0] = a;
                  t.alleles[0] = reinterpret_cast<uint8_t const *>(g.dna)[g.var_dna[fv + a]];
                  t.site[0] = site;
                  t.seg_base[1] = g.ref_dna[site + 1] - 1;
                  n = 1;
                  uint32_t const next_len = g.ref_len[site + 1];
                  done = next_len >= tail_len - 1;
                  at = 1 + next_len;
                  r = site + 1;
                }
                if (n == 0)
                  r = INVALID;
              }
              while (!done && n < E4::TS && r != INVALID && r + 1 < g.n_ref)
              {
                uint32_t const nv = g.ref_nvar[r], fv = g.ref_first_var[r];
                bool snp = nv >= 2 && nv <= 4;
                uint32_t codes = 0;
                for (uint32_t a = 0; a < 4 && snp; ++a)
                  if (a < nv)
                  {
                    snp = g.var_len[fv + a] == 1;
                    if (snp)
                      codes |= static_cast<uint32_t>(reinterpret_cast<uint8_t const *>(g.dna)[g.var_dna[fv + a]]) << (8 * a);
                  }
                if (!snp)
                {
                  if constexpr (E4::INDEL_TAIL)
                  if (n == 0 && nv >= 2 && nv <= EXPRESS4_INDEL_ALLELES)
                  {
                    // alleles of any length: every one has to end inside the tail with at least one character left for
                    // the next reference node, and that node has to hold the rest (else the walk ends inside an allele
                    // or runs over a second site: pass 2)
                    bool fits = true;
                    uint32_t const next_len = g.ref_len[r + 1];
                    for (uint32_t a = 0; a < EXPRESS4_INDEL_ALLELES; ++a)
                      if (a < nv)
                      {
                        uint32_t const vl = g.var_len[fv + a];
                        // (an allele the tail ends in -- or at the end of -- is compared as far as the tail goes)
                        fits = fits && (at + vl >= tail_len || tail_len - at - vl <= next_len);
                        ws.xtail[gi].alen[a] = vl;
                        ws.xtail[gi].aoff[a] = g.var_dna[fv + a];
                        ws.xtail[gi].amm[a] = 0;
                      }
                    GTX_E4_NOTE(!fits, 13); // behind an allele of the indel site the tail runs over the next site
                    if (fits)
                    {
                      t.x.indel = 1;
                      t.x.cmp_len = at;
                      ws.xtail[gi].next_off = g.ref_dna[r + 1];
                      ws.xtail[gi].next_order = g.ref_order[r + 1];
                      ws.xtail[gi].site_order = g.ref_order[r] + g.ref_len[r];
                      t.site[0] = r;
                      t.nall[0] = nv;
                      done = true;
                    }
                  }
                  break;
                }
                t.site_at[n] = at;
                t.nall[n] = nv;
                t.afirst[n] = 0;
                t.alleles[n] = codes;
                t.site[n] = r;
                t.seg_base[n + 1] = g.ref_dna[r + 1] - (at + 1);
                ++n;
                uint32_t const next_len = g.ref_len[r + 1];
                if (next_len >= tail_len - at - 1)
                  done = true;
                at += 1 + next_len;
                ++r;
              }
              if (done)
              {
                t.ok = 1;
                t.tail_len = tail_len;
                t.nsite = n;
                if constexpr (E4::INDEL_TAIL)
                  if (!t.x.indel)
                    t.x.cmp_len = tail_len;
              }
              else
                for (uint32_t k = 0; k < E4::TS; ++k)
                  t.site_at[k] = INVALID;
            }
          }
        }
      }
      GTX_E4_NOTE(seeded && !t.ok && t.prs != 0 && t.head_len == 0, 8); // the walk at the read's start leaves the node
      GTX_E4_NOTE(seeded && !t.ok && !(t.prs != 0 && t.head_len == 0), 9); // the walk at the read's end is not simple
      ws.tail[gi] = t;
    }
  });
  W::lds_sync();
  GTX_PROF_TICK(5)

  // ---- the two compares, 16 characters per group and round (count_mismatches[_backward], graph_utils.hpp:7-69)
  PU got_l, hgot_l;
  PB killed_l, hkilled_l;
  W::lanes([&](uint32_t l) {
    got_l[l] = 0;
    hgot_l[l] = 0;
    killed_l[l] = false;
    hkilled_l[l] = false;
  });
  for (uint32_t r = 0; r < AlignCfg::MAX_READ / 16; ++r)
  {
    PB k_l, x_l, hk_l, hx_l, any_l;
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4, i = 16 * r + (l & 15u);
      Express4Tail<E4> const & t = ws.tail[gi];
      bool k = false, x = false, hk = false, hx = false;
      bool const on = seeded_l[l] && t.ok;
      uint32_t base = t.seg_base[0];
      bool on_site = false; // (the characters over variant sites: verdict below)
      for (uint32_t sk = 0; sk < E4::TS; ++sk)
      {
        uint32_t const sa = t.site_at[sk];
        on_site = on_site || i == sa;
        if (sa != INVALID && i > sa)
          base = t.seg_base[sk + 1];
      }
      uint32_t cmp_len = t.tail_len; // (an indel tail: only the characters in front of the site)
      if constexpr (E4::INDEL_TAIL)
        cmp_len = t.x.cmp_len;
      if (on && i < cmp_len && !on_site)
      {
        uint8_t const gc = reinterpret_cast<uint8_t const *>(g.dna)[base + i];
        uint8_t const rc = ws.s[gi].rd[t.pre + i];
        k = gc == DNA_KILL;
        x = gc != rc && rc != 15 && gc != 15;
      }
      if (on && i < t.head_len)
      {
        uint8_t const gc = reinterpret_cast<uint8_t const *>(g.dna)[t.head_off - i];
        uint8_t const rc = ws.s[gi].rd[t.prs - i];
        hk = gc == DNA_KILL;
        hx = gc != rc && rc != 15 && gc != 15;
      }
      k_l[l] = k;
      x_l[l] = x;
      hk_l[l] = hk;
      hx_l[l] = hx;
      any_l[l] = on && (16 * r < t.tail_len || 16 * r < t.head_len);
    });
    if (W::ballot(any_l) == 0)
      break;
    uint64_t const KILL = W::ballot(k_l), X = W::ballot(x_l), HKILL = W::ballot(hk_l), HX = W::ballot(hx_l);
    W::lanes([&](uint32_t l) {
      uint32_t const sh = 16 * (l >> 4);
      got_l[l] = got_l[l] + static_cast<uint32_t>(__builtin_popcount(static_cast<uint32_t>(X >> sh) & 0xFFFFu));
      hgot_l[l] = hgot_l[l] + static_cast<uint32_t>(__builtin_popcount(static_cast<uint32_t>(HX >> sh) & 0xFFFFu));
      killed_l[l] = killed_l[l] || (static_cast<uint32_t>(KILL >> sh) & 0xFFFFu) != 0;
      hkilled_l[l] = hkilled_l[l] || (static_cast<uint32_t>(HKILL >> sh) & 0xFFFFu) != 0;
    });
  }

  GTX_PROF_TICK(6)
  // ---- tails over an indel site: the rest of the tail against every allele's candidate (allele bases, then the next
  //      reference node), 16 characters per group and round
  if constexpr (E4::INDEL_TAIL)
  {
    PB ind_l;
    W::lanes([&](uint32_t l) {
      Express4Tail<E4> const & t = ws.tail[l >> 4];
      ind_l[l] = seeded_l[l] && t.ok && t.x.indel;
    });
    if (W::ballot(ind_l) != 0)
      for (uint32_t a = 0; a < EXPRESS4_INDEL_ALLELES; ++a)
        for (uint32_t r = 0; r < AlignCfg::MAX_READ / 16; ++r)
        {
          PB x_l, any_l;
          W::lanes([&](uint32_t l) {
            uint32_t const gi = l >> 4;
            Express4Tail<E4> const & t = ws.tail[gi];
            bool const on = ind_l[l] && a < t.nall[0];
            uint32_t const i = t.x.cmp_len + 16 * r + (l & 15u);
            bool x = false;
            if (on && i < t.tail_len)
            {
              Express4IndelAlleles<E4::INDEL_TAIL> const & xa = ws.xtail[gi];
              uint32_t const k = i - t.x.cmp_len, al = xa.alen[a];
              uint8_t const gc = reinterpret_cast<uint8_t const *>(g.dna)[k < al ? xa.aoff[a] + k : xa.next_off + (k - al)];
              uint8_t const rc = ws.s[gi].rd[t.pre + i];
              x = gc != rc && rc != 15 && gc != 15;
            }
            x_l[l] = x;
            any_l[l] = on && t.x.cmp_len + 16 * r < t.tail_len;
          });
          if (W::ballot(any_l) == 0)
            break;
          uint64_t const XA = W::ballot(x_l);
          W::lanes([&](uint32_t l) {
            if ((l & 15u) == 0 && ind_l[l])
              ws.xtail[l >> 4].amm[a] += static_cast<uint32_t>(__builtin_popcount(static_cast<uint32_t>(XA >> (16 * (l >> 4))) & 0xFFFFu));
          });
        }
  }

  GTX_PROF_TICK(7)
  // ---- verdict and record (leader lanes)
  PB fail_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4;
    bool fail = pass2_l[l];
    if (alive_l[l])
    {
      Express4Tail<E4> const & t = ws.tail[gi];
      fail = !(seeded_l[l] && t.ok);
      if (!fail && (l & 15u) == 0)
      {
        SeedWorkspace const & s = ws.s[gi];
        uint32_t const L = len_l[l], lo = lo_l[l], hi = hi_l[l];
        uint32_t start = s.fs_start[lo], end = s.fs_end[hi], rs = t.prs, re = t.pre, mism = mism_l[l];
        if (t.head_len)
        {
          uint32_t const budget = 2 + t.head_len / 11 < 7 ? 2 + t.head_len / 11 : 7; // genotype_paths.cpp:571-577
          if (!hkilled_l[l] && hgot_l[l] <= budget)
          {
            start -= t.head_len - 1;
            rs = 0;
            mism += hgot_l[l];
          }
        }
        uint32_t tail_sites = 0;
        uint32_t tail_mask[E4::TS];
        if (t.tail_len)
        {
          uint32_t const budget = 2 + t.tail_len / 11 < 7 ? 2 + t.tail_len / 11 : 7; // genotype_paths.cpp:505-511
          uint32_t got = got_l[l];
          bool killed = killed_l[l];
          for (uint32_t sk = 0; sk < E4::TS; ++sk) // the characters over the variant sites: the alleles that mismatch least
          {
            tail_mask[sk] = 0;
            if (sk < t.nsite)
            {
              uint8_t const rc = s.rd[t.pre + t.site_at[sk]];
              uint32_t best = 2;
              for (uint32_t a = 0; a < 4; ++a)
                if (a < t.nall[sk])
                {
                  uint8_t const gc = static_cast<uint8_t>(t.alleles[sk] >> (8 * a));
                  killed = killed || gc == DNA_KILL;
                  uint32_t const xa = (gc != rc && rc != 15 && gc != 15) ? 1u : 0u;
                  if (xa < best)
                  {
                    best = xa;
                    tail_mask[sk] = 0;
                  }
                  if (xa == best)
                    tail_mask[sk] |= 1u << a;
                }
              got += best;
            }
          }
          bool tie = false;
          uint32_t indel_end = 0, indel_mask = 0;
          bool indel = false;
          if constexpr (E4::INDEL_TAIL)
            indel = t.x.indel != 0;
          if constexpr (E4::INDEL_TAIL)
            if (indel)
            {
              // One candidate per allele; the labels of the best ones are the walk's result.  A candidate's last
              // character lies in the reference node behind the site, or inside the allele (then a position of the
              // variant node, special beyond the reference allele's reach: graph.cpp:1232-1243).  Best candidates
              // that end at the same position are one label list with equal ends, i.e. one path with all their
              // alleles; different ends would be several paths (pass 2).
              uint32_t best = INVALID;
              for (uint32_t a = 0; a < EXPRESS4_INDEL_ALLELES; ++a)
                if (a < t.nall[0] && ws.xtail[gi].amm[a] < best)
                  best = ws.xtail[gi].amm[a];
              got += best;
              uint32_t const beyond = t.tail_len - t.x.cmp_len;
              for (uint32_t a = 0; a < EXPRESS4_INDEL_ALLELES; ++a)
                if (a < t.nall[0] && ws.xtail[gi].amm[a] == best)
                {
                  uint32_t const al = ws.xtail[gi].alen[a];
                  uint32_t const e = beyond > al ? ws.xtail[gi].next_order + (beyond - al) - 1
                                                 : g_special_of(g, t.site[0], ws.xtail[gi].site_order + beyond - 1);
                  tie = tie || (indel_mask != 0 && e != indel_end);
                  indel_end = e;
                  indel_mask |= 1u << a;
                }
              tie = tie && got <= budget;
            }
          if (!killed && got <= budget)
          {
            re = L - 1;
            mism += got;
            if (indel)
            {
              end = indel_end;
              tail_sites = 1;
              tail_mask[0] = indel_mask;
            }
            else
            {
              end += t.tail_len - 1;
              tail_sites = t.nsite;
            }
          }
          GTX_E4_NOTE(tie, 12); // the best alleles of an indel site end at different positions: several paths
          fail = fail || tie;
        }
        uint32_t longest = re - rs + 1;
        // variant sites of the path: every merge puts the new label's site in front, intersecting the allele sets when
        // the path already carries the site (path.cpp:38-82); an empty intersection makes the merge fail (declined)
        uint32_t * vs = ws.vsite[gi]; // (in LDS: indexed arrays in registers would spill)
        uint64_t * vm = ws.vmask[gi];
        uint32_t nvar = 0;
        uint32_t const var_run = static_cast<uint32_t>((VAR >> (16 * gi)) & 0xFFFFu) & (((2u << hi) - 1u) & ~((1u << lo) - 1u));
        bool clash = false;
        auto push_front = [&](uint32_t site, uint64_t mask)
        {
          uint32_t k = 0;
          while (k < nvar && vs[k] != site)
            ++k;
          if (k < nvar)
          {
            mask &= vm[k];
            clash = clash || mask == 0;
          }
          else if (nvar < E4::VS_CAP)
            ++nvar;
          else
          {
            clash = true; // more sites than the list holds: pass 2
            return;
          }
          for (; k > 0; --k)
          {
            vs[k] = vs[k - 1];
            vm[k] = vm[k - 1];
          }
          vs[0] = site;
          vm[0] = mask;
        };
        if (var_run != 0 || tail_sites != 0) // (most reads of a sparse graph carry no variant: skip all of this)
        {
          // a k-mer's (or the tail's) sites keep their label order in front of the older ones: pushed last to first
          for (uint32_t k = lo; k <= hi; ++k)
            if ((var_run >> k) & 1u)
              for (uint32_t i = ws.kn[gi][k]; i-- > 0;)
                push_front(ws.ksite[gi][k][i], ws.kmask[gi][k][i]);
          for (uint32_t sk = E4::TS; sk-- > 0;)
            if (sk < tail_sites)
              push_front(t.site[sk], static_cast<uint64_t>(tail_mask[sk]) << t.afirst[sk]);
        }
        bool const with_var = nvar != 0;
        GTX_E4_NOTE(clash || 6 + 3 * nvar > rec_words, 10); // allele sets do not intersect / too many sites
        if (clash || 6 + 3 * nvar > rec_words)
        {
          fail = true; // (nothing written: pass 2 redoes the task)
          nvar = 0;
        }
        uint32_t np = 1;
        if (mism > 10) // remove_paths_with_too_many_mismatches on one path
        {
          np = 0;
          longest = 0;
        }
        uint32_t * rec = records + static_cast<uint64_t>(read_l[l]) * 2 * rec_words;
        if (!fail)
        {
          rec[0] = np;
          rec[1] = longest | (L << 16) | ((np && with_var) ? GTX_REC_HAS_VARIANTS : 0u);
          if (np)
          {
            rec[2] = start;
            rec[3] = end;
            rec[4] = rs | (re << 16);
            rec[5] = mism | (nvar << 16);
            for (uint32_t k = 0; k < nvar; ++k)
            {
              rec[6 + 3 * k] = vs[k];
              rec[7 + 3 * k] = static_cast<uint32_t>(vm[k]);
              rec[8 + 3 * k] = static_cast<uint32_t>(vm[k] >> 32);
            }
          }
        }
      }
    }
    fail_l[l] = fail;
  });
  uint64_t const FAIL = W::ballot(fail_l);
  W::lds_sync();
  GTX_PROF_TICK(8)
#ifdef GTX_PROF
  GTX_LEAD ws.prof_acc[15] += 1;
#endif
  return (FAIL & 1ull ? 1u : 0u) | (FAIL >> 16 & 1ull ? 2u : 0u) | (FAIL >> 32 & 1ull ? 4u : 0u) | (FAIL >> 48 & 1ull ? 8u : 0u);
}
