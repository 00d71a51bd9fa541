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

struct Express4Tail // handed from a group's leader lane to its 16 lanes
{
  uint32_t dna_off, tail_len, pre; // walk at the read's end: arena offset of the path's last base, characters, read offset
  uint32_t head_off, head_len, prs; // walk at the read's start (backwards from the path's first base)
  uint32_t ok;
  // a tail that runs over one SNP-like site (every allele one base): characters [0, room) in the node the path ends in,
  // character `room` against the alleles, the rest in the next reference node
  uint32_t room, nall, alleles, next_off, site; // alleles: one comparison code per byte
};

struct Express4Workspace
{
  SeedWorkspace s[4];
  Express4Tail tail[4];
  uint32_t ksite[4][AlignCfg::KC], kallele[4][AlignCfg::KC]; // the variant (site, allele) of every k-mer's label
  uint32_t vsite[4][AlignCfg::KC + 1];                       // the path's sites in record order (built by the leader lane)
  uint64_t vmask[4][AlignCfg::KC + 1];
};

// Returns a 4-bit mask: bit gi set = task first + gi must go through pass 2.
template <class W>
GTX_DEV uint32_t express4(GraphView const & g, IndexView const & ix, Express4Workspace & ws, uint8_t const * seq, uint32_t seq_stride,
                          gtx_read_meta const * meta, uint32_t first, uint32_t n_valid, uint32_t * records, uint32_t rec_words,
                          bool decline_all = false)
{
  using PB = typename W::template PerLane<bool>;
  using PU = typename W::template PerLane<uint32_t>;
  constexpr uint32_t KC = AlignCfg::KC, HE_CAP = AlignCfg::HE_CAP;
  bool const use_halves = ix.half_bucket_cap != 0;

  // ---- per group: the read, whether it is a candidate at all
  PB alive_l, pass2_l;
  PU len_l, nk_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4;
    bool const valid = gi < n_valid;
    uint32_t const len = valid ? static_cast<uint32_t>(meta[first + gi].l_qseq) : 0u;
    bool const too_short = len < 2 * K - 1, too_long = len > AlignCfg::MAX_READ;
    uint32_t const n_k = len < K ? 0 : 1 + (len - K) / (K - 1);
    if (valid && (too_short || too_long) && (l & 15u) == 0)
    {
      // align_read (alignment.cpp:331-363): reads shorter than 2K-1 stay unaligned
      uint32_t * rec = records + static_cast<uint64_t>(first + gi) * 2 * rec_words;
      rec[0] = too_long ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
      rec[1] = len << 16;
    }
    bool const candidate = valid && !too_short && !too_long;
    bool const fits = candidate && n_k <= KC && use_halves && rec_words >= 9 && !decline_all; // (decline_all: test switch)
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

  // ---- unpack the reads (BAM nibbles -> one code per byte), 4 bases per lane and round
  for (uint32_t it = 0; it < AlignCfg::MAX_READ / 64; ++it)
    W::lanes([&](uint32_t l) {
      uint32_t const gi = l >> 4, word = it * 16 + (l & 15u), len = len_l[l];
      if (alive_l[l] && 4 * word < len)
      {
        uint8_t const * seq4 = seq + static_cast<uint64_t>(first + gi) * seq_stride;
        uint32_t packed = 0;
        for (uint32_t k = 0; k < 4; ++k)
        {
          uint32_t const i = 4 * word + k;
          uint32_t c = 15;
          if (i < len)
          {
            c = (seq4[i >> 1] >> ((~i & 1u) << 2)) & 15u;
            if (c == 0)
              c = 15; // '=' assigned to a seqan Iupac becomes N (hts_parallel_reader.cpp:226-243)
          }
          packed |= c << (8 * k);
        }
        reinterpret_cast<uint32_t *>(ws.s[gi].rd)[word] = packed;
      }
    });
  W::lds_sync();

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
        uint32_t off, cnt;
        IndexSlot const * hit;
        bucket_find(w == 0 ? ix.slots : ix.hslots, w == 0 ? ix.log2_cap : ix.h_log2_cap, w == 0 ? q : half_key(q, w - 1), off, cnt, &hit);
        if (w == 0)
        {
          s.off0[i] = off;
          s.cnt0[i] = cnt;
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
      }
    });
  }
  W::lds_sync();

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
  }

  // ---- fast seeding, lane j < n_k of a group = k-mer j (the rules and their justification: seed_stage).
  //      One k-mer of the read may have no label at all (two or more errors, an error next to an N): a "hole".
  PB bad_l, var_l, mm_l, hole_l, par_l; // par: the k-mer also starts a parallel (+1 mismatch) chain when it opens a run
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4, j = l & 15u;
    SeedWorkspace & s = ws.s[gi];
    bool bad = false, has_var = false, mm = false, hole = false, par = false;
    if (alive_l[l] && j < nk_l[l])
    {
      uint32_t const c0 = s.cnt0[j];
      if (s.nkeys0[j] != 1)
      {
        uint32_t const a0 = j < 4 ? s.acnt[j][0] : 0xFFFFFFFFu, a1 = j < 4 ? s.acnt[j][1] : 0u, a2 = j < 4 ? s.acnt[j][2] : 0u,
                       a3 = j < 4 ? s.acnt[j][3] : 0u;
        hole = a0 == 0 && a1 + a2 + a3 == 0;
        bad = !hole && (a0 > 1 || a0 + a1 + a2 + a3 != 1);
        par = true; // a multi-key list is added twice (0 and 1 mismatches)
        if (!bad && !hole)
        {
          DevLabel const lb = s.xl[j][0];
          bad = lb.site != INVALID;
          s.fs_start[j] = lb.start;
          s.fs_end[j] = lb.end;
        }
      }
      else if (!(bad = c0 > 1 || s.hcnt[j][0] > HE_CAP || s.hcnt[j][1] > HE_CAP))
      {
        uint64_t const q = s.key0[j];
        uint32_t nb = 0, nb_off = 0;
        for (uint32_t side = 0; side < 2; ++side)
          for (uint32_t e = 0; e < s.hcnt[j][side]; ++e)
          {
            HalfEntry const & he = s.he[j][side][e];
            uint32_t nn;
            if (hamming1_neighbour(he.key, q, nn))
            {
              nb += he.cnt;
              nb_off = he.off;
            }
          }
        if (c0 + nb == 0)
          hole = true;
        else if (c0 + nb == 1)
        {
          DevLabel const lb = c0 ? s.xl[j][0] : ix.labels[nb_off];
          mm = c0 == 0;
          s.fs_start[j] = lb.start;
          s.fs_end[j] = lb.end;
          if (lb.site != INVALID) // the only label there is lies on a variant (e.g. an error inside a k-mer over a SNP)
          {
            bad = g.is_sv_graph != 0;
            has_var = !bad;
            ws.ksite[gi][j] = lb.site;
            ws.kallele[gi][j] = lb.allele;
          }
        }
        else if (c0 == 1 && nb <= 3 && !g.is_sv_graph)
        {
          DevLabel const lb = s.xl[j][0];
          bad = lb.site == INVALID;
          for (uint32_t side = 0; side < 2 && !bad; ++side)
            for (uint32_t e = 0; e < s.hcnt[j][side]; ++e)
            {
              HalfEntry const & he = s.he[j][side][e];
              uint32_t nn;
              if (hamming1_neighbour(he.key, q, nn))
                for (uint32_t k = 0; k < he.cnt; ++k)
                {
                  DevLabel const nl = ix.labels[he.off + k];
                  bad = bad || nl.start != lb.start || nl.end != lb.end || nl.site != lb.site;
                }
            }
          if (!bad)
          {
            has_var = true;
            par = true; // the site's other alleles start chains with one more mismatch
            s.fs_start[j] = lb.start;
            s.fs_end[j] = lb.end;
            ws.ksite[gi][j] = lb.site;
            ws.kallele[gi][j] = lb.allele;
          }
        }
        else
          bad = true;
      }
    }
    bad_l[l] = bad;
    var_l[l] = has_var;
    mm_l[l] = mm;
    hole_l[l] = hole;
    par_l[l] = par;
  });
  uint64_t const BAD = W::ballot(bad_l), VAR = W::ballot(var_l), MM = W::ballot(mm_l), HOLE = W::ballot(hole_l), PAR = W::ballot(par_l);
  W::lds_sync();
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
      uint32_t const h = static_cast<uint32_t>(__builtin_ctz(hole));
      uint32_t const left = h, right = n_k - 1 - h; // k-mers on either side
      ok = (hole & (hole - 1u)) == 0 && left != right;
      if (left > right)
        hi = h - 1;
      else
        lo = h + 1;
      // A run that starts (after the hole) with a multi-key k-mer, or with a k-mer on a variant whose other alleles are
      // indexed, starts with parallel chains (the list's +1-mismatch copy, the other alleles); the start walk then yields
      // one label list per chain, the later lists find no chain left to merge with, become paths of their own and are
      // walked to duplicates of the result (the reference really returns the path twice): left to pass 2.
      if (ok && lo > 0 && ((static_cast<uint32_t>(PAR >> sh) >> lo) & 1u))
        ok = false;
    }
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
    seeded_l[l] = seeded;
    uint32_t const lo = lo_l[l], hi = hi_l[l];
    uint32_t const run_mask = seeded ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    mism_l[l] = static_cast<uint32_t>(__builtin_popcount(mm & run_mask));
    if ((l & 15u) == 0)
    {
      Express4Tail t{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
            if (w != INVALID && (w & 255u) >= tail_len)
            {
              t.ok = 1;
              t.dna_off = w >> 8;
              t.tail_len = tail_len;
              t.room = tail_len;
            }
            else if (w != INVALID && (w & 255u) < 255u && !g.is_sv_graph && g.pos_node)
            {
              // The tail leaves the node over a variant site.  When every allele of that site is a single base (a SNP)
              // and the rest fits in the next reference node, Graph::get_labels_forward has one candidate per allele,
              // they differ in that one character, and the labels of the best ones share (start, end): one path whose
              // allele set is the best alleles (make_pp unites labels with equal ends).
              uint32_t const r = g.pos_node[anchor - g.first_order], room = w & 255u;
              if (r != INVALID && r + 1 < g.n_ref)
              {
                uint32_t const nv = g.ref_nvar[r], fv = g.ref_first_var[r];
                uint32_t const rest = tail_len - room - 1;
                bool snp = nv >= 2 && nv <= 4 && g.ref_len[r + 1] >= rest;
                uint32_t codes = 0;
                for (uint32_t a = 0; a < 4 && snp; ++a)
                  if (a < nv)
                  {
                    snp = g.var_len[fv + a] == 1;
                    if (snp)
                      codes |= static_cast<uint32_t>(reinterpret_cast<uint8_t const *>(g.dna)[g.var_dna[fv + a]]) << (8 * a);
                  }
                if (snp)
                {
                  t.ok = 1;
                  t.dna_off = w >> 8;
                  t.tail_len = tail_len;
                  t.room = room;
                  t.nall = nv;
                  t.alleles = codes;
                  t.next_off = g.ref_dna[r + 1];
                  t.site = r;
                }
              }
            }
          }
        }
      }
      ws.tail[gi] = t;
    }
  });
  W::lds_sync();

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
      Express4Tail const t = ws.tail[gi];
      bool k = false, x = false, hk = false, hx = false;
      bool const on = seeded_l[l] && t.ok;
      if (on && i < t.tail_len && i != t.room) // (character `room`, if inside the tail, is the variant: verdict below)
      {
        uint8_t const gc = reinterpret_cast<uint8_t const *>(g.dna)[i < t.room ? t.dna_off + i : t.next_off + (i - t.room - 1)];
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

  // ---- verdict and record (leader lanes)
  PB fail_l;
  W::lanes([&](uint32_t l) {
    uint32_t const gi = l >> 4;
    bool fail = pass2_l[l];
    if (alive_l[l])
    {
      Express4Tail const t = ws.tail[gi];
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
        uint32_t tail_site = INVALID, tail_mask = 0;
        if (t.tail_len)
        {
          uint32_t const budget = 2 + t.tail_len / 11 < 7 ? 2 + t.tail_len / 11 : 7; // genotype_paths.cpp:505-511
          uint32_t got = got_l[l];
          bool killed = killed_l[l];
          if (t.nall) // the character over the variant site: the alleles that mismatch least
          {
            uint8_t const rc = s.rd[t.pre + t.room];
            uint32_t best = 2;
            for (uint32_t a = 0; a < t.nall; ++a)
            {
              uint8_t const gc = static_cast<uint8_t>(t.alleles >> (8 * a));
              killed = killed || gc == DNA_KILL;
              uint32_t const xa = (gc != rc && rc != 15 && gc != 15) ? 1u : 0u;
              if (xa < best)
              {
                best = xa;
                tail_mask = 0;
              }
              if (xa == best)
                tail_mask |= 1u << a;
            }
            got += best;
          }
          if (!killed && got <= budget)
          {
            end += t.tail_len - 1;
            re = L - 1;
            mism += got;
            if (t.nall)
              tail_site = t.site;
          }
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
          else
            ++nvar;
          for (; k > 0; --k)
          {
            vs[k] = vs[k - 1];
            vm[k] = vm[k - 1];
          }
          vs[0] = site;
          vm[0] = mask;
        };
        if (var_run != 0 || tail_site != INVALID) // (most reads of a sparse graph carry no variant: skip all of this)
        {
          for (uint32_t k = lo; k <= hi; ++k)
            if ((var_run >> k) & 1u)
              push_front(ws.ksite[gi][k], 1ull << ws.kallele[gi][k]);
          if (tail_site != INVALID)
            push_front(tail_site, tail_mask);
        }
        bool const with_var = nvar != 0;
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
        uint32_t * rec = records + static_cast<uint64_t>(first + gi) * 2 * rec_words;
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
  return (FAIL & 1ull ? 1u : 0u) | (FAIL >> 16 & 1ull ? 2u : 0u) | (FAIL >> 32 & 1ull ? 4u : 0u) | (FAIL >> 48 & 1ull ? 8u : 0u);
}
