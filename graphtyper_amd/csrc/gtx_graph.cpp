// gtx_graph.cpp -- host graph builder: variant records + reference sequence -> the SoA node tables of gtx_graph_view.
//
// Behaviour contract: Graph::add_genomic_region (src/graph/graph.cpp:41-339) with VarRecord::merge / merge_one_path /
// merge_all (src/graph/var_record.cpp:179-371), make_alt / is_ok_to_merge_alts (src/graph/alt.cpp:60-141) and the
// constructor's prefix extension (src/graph/genomic_region.cpp:236-256).  Data model of this implementation: a site is a
// vector of alleles with allele 0 = REF, so operations that touch "the reference and every alt" are one loop; event sets
// are small sorted vectors.  Output order (alts sorted by sequence, graph.cpp:290-293) makes the node tables unique.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "gtx_ctx.hpp"

namespace
{
using Events = std::vector<int64_t>; // sorted, unique

void ev_add(Events & dst, Events const & src)
{
  Events out;
  out.reserve(dst.size() + src.size());
  std::set_union(dst.begin(), dst.end(), src.begin(), src.end(), std::back_inserter(out));
  dst.swap(out);
}

bool ev_has(Events const & e, int64_t x)
{
  return std::binary_search(e.begin(), e.end(), x);
}

struct Allele
{
  std::string seq;
  Events events, anti;
};

struct Site
{
  uint32_t pos = 0;            // 0-based
  std::vector<Allele> alleles; // [0] = REF
  bool is_sv = false;
  bool dead = false; // merged into the next site

  std::string const & ref() const { return alleles[0].seq; }
  size_t n_alts() const { return alleles.size() - 1; }
  long end() const { return static_cast<long>(pos) + static_cast<long>(ref().size()); }

  bool snps_only() const // VarRecord::is_snp_or_snps
  {
    for (size_t a = 1; a < alleles.size(); ++a)
      if (alleles[a].seq.size() != ref().size())
        return false;
    return true;
  }

  bool any_longer_than(size_t n) const // VarRecord::is_any_seq_larger_than
  {
    for (auto const & a : alleles)
      if (a.seq.size() > n)
        return true;
    return false;
  }
};

// alts of `from` (index >= first) that are not already among the first `n_before` alts of `to` are appended
// (move_alts, var_record.cpp:81-107: only alts present before the call count as duplicates)
void adopt_alts(Site & to, size_t n_before, std::vector<Allele> & from, size_t first)
{
  for (size_t i = first; i < from.size(); ++i)
  {
    bool dup = false;
    for (size_t a = 1; a < 1 + n_before && !dup; ++a)
      dup = to.alleles[a].seq == from[i].seq;
    if (!dup)
      to.alleles.push_back(std::move(from[i]));
  }
}

// bring both sites onto the same reference span starting at prev.pos: `cur` gets prev's leading bases, then whichever
// reference is shorter is padded with the other's tail (insert_prior_sequence + extend_smaller_record)
void align_spans(Site & cur, Site & prev)
{
  if (prev.pos < cur.pos)
  {
    std::string const lead = prev.ref().substr(0, cur.pos - prev.pos);
    for (auto & a : cur.alleles)
      a.seq.insert(0, lead);
    cur.pos = prev.pos;
  }
  Site & shorter = cur.ref().size() < prev.ref().size() ? cur : prev;
  Site const & longer = cur.ref().size() < prev.ref().size() ? prev : cur;
  if (shorter.ref().size() != longer.ref().size())
  {
    std::string const tail = longer.ref().substr(shorter.ref().size());
    for (auto & a : shorter.alleles)
      a.seq += tail;
  }
}

void inherit_ref_events(Site & cur, Site const & prev)
{
  for (auto & a : cur.alleles)
  {
    ev_add(a.events, prev.alleles[0].events);
    ev_add(a.anti, prev.alleles[0].anti);
  }
}

bool may_join(Allele const & before, Allele const & after) // is_ok_to_merge_alts
{
  for (int64_t e : after.events)
    if (e >= 0 && ev_has(before.anti, e))
      return false;
  return true;
}

Allele joined(Allele const & before, Allele const & after, size_t skip) // make_alt
{
  Allele n(before);
  n.seq.append(after.seq, skip, std::string::npos);
  ev_add(n.events, after.events);
  ev_add(n.anti, after.anti);
  return n;
}

void merge_one_path(Site & cur, Site & prev) // var_record.cpp:179-200
{
  align_spans(cur, prev);
  size_t const before = cur.n_alts();
  inherit_ref_events(cur, prev);
  adopt_alts(cur, before, prev.alleles, 1);
}

void merge_overlapping(Site & cur, Site & prev, long extra_suffix) // VarRecord::merge, var_record.cpp:270-371
{
  long const jump = static_cast<long>(cur.pos) - static_cast<long>(prev.pos);
  long const cur_ref0 = static_cast<long>(cur.ref().size());
  long const cur_ref1 = cur_ref0 + std::max<long>(jump, 0); // after the lead was prepended
  align_spans(cur, prev);
  long const grown = static_cast<long>(cur.ref().size()) - cur_ref1;
  std::vector<Allele> combos;
  for (size_t p = 1; p < prev.alleles.size(); ++p)
  {
    Allele const & pa = prev.alleles[p];
    if (static_cast<long>(pa.seq.size()) <= cur_ref0)
      continue;
    long const offset = static_cast<long>(cur.ref().size()) - static_cast<long>(pa.seq.size());
    if (jump - offset < 0)
      continue;
    long same_tail = 0;
    for (long k = 0, n = static_cast<long>(std::min(cur.ref().size(), pa.seq.size())); k < n; ++k, ++same_tail)
      if (cur.ref()[cur.ref().size() - 1 - k] != pa.seq[pa.seq.size() - 1 - k])
        break;
    if (same_tail < grown + extra_suffix)
      continue;
    Allele head(pa);
    head.seq.resize(jump - offset);
    for (size_t c = 1; c < cur.alleles.size(); ++c)
      if (may_join(head, cur.alleles[c]))
        combos.push_back(joined(head, cur.alleles[c], static_cast<size_t>(jump)));
  }
  inherit_ref_events(cur, prev);
  // drop previous alts that are anti to an event the (merged) reference allele now carries
  std::vector<Allele> kept;
  kept.push_back(Allele());
  for (size_t p = 1; p < prev.alleles.size(); ++p)
  {
    bool anti_ref = false;
    for (int64_t a : prev.alleles[p].anti)
      anti_ref = anti_ref || ev_has(cur.alleles[0].events, a);
    if (!anti_ref)
      kept.push_back(std::move(prev.alleles[p]));
  }
  size_t before = cur.n_alts();
  adopt_alts(cur, before, kept, 1);
  before = cur.n_alts();
  adopt_alts(cur, before, combos, 0);
}

void merge_adjacent(Site & cur, Site & prev) // VarRecord::merge_all, var_record.cpp:202-268
{
  if (prev.end() != static_cast<long>(cur.pos))
  {
    merge_overlapping(cur, prev, 0);
    return;
  }
  std::vector<Allele> combos;
  for (size_t p = 1; p < prev.alleles.size(); ++p)
  {
    for (size_t c = 1; c < cur.alleles.size(); ++c)
      if (may_join(prev.alleles[p], cur.alleles[c]))
        combos.push_back(joined(prev.alleles[p], cur.alleles[c], 0));
    combos.push_back(joined(prev.alleles[p], cur.alleles[0], 0)); // previous alt + current reference
  }
  for (auto & a : cur.alleles) // reference included: R + S, R + C, ...
    a.seq.insert(0, prev.ref());
  inherit_ref_events(cur, prev);
  cur.pos = prev.pos;
  size_t const before = cur.n_alts();
  adopt_alts(cur, before, combos, 0);
  // an allele that carries an event it is itself anti to cannot exist
  for (size_t a = cur.alleles.size(); a-- > 1;)
  {
    bool self_anti = false;
    for (int64_t x : cur.alleles[a].anti)
      self_anti = self_anti || ev_has(cur.alleles[a].events, x);
    if (self_anti)
      cur.alleles.erase(cur.alleles.begin() + a);
  }
}

bool prefix_related(std::string const & a, std::string const & b)
{
  size_t const n = std::min(a.size(), b.size());
  return a.compare(0, n, b, 0, n) == 0;
}

constexpr size_t MAX_HAPS = 2560; // MAX_NUMBER_OF_HAPLOTYPES

} // namespace

struct gtx_graph
{
  std::vector<uint32_t> ref_order, ref_len, ref_dna_off, ref_nvar, ref_first_var;
  std::vector<uint32_t> var_order, var_len, var_dna_off, var_out_ref, event_off;
  std::vector<int64_t> event_val;
  std::string dna;
  std::string sv_table; // Graph::SVs of a graph made from files with structural variants (gtx_graph_sv_table)
};

namespace gtx
{
void graph_set_sv_table(gtx_graph * g, std::string table) { g->sv_table = std::move(table); }
} // namespace gtx

extern "C"
{
  int gtx_graph_build(const char * reference, uint64_t reference_len, int64_t region_begin, int64_t region_end,
                      const gtx_record * records, uint32_t n_records, int add_all_variants, int is_sv_graph, int extend_prefix,
                      gtx_graph ** out)
  {
    using namespace gtx;
    if (!reference || !out || (n_records && !records))
    {
      g_last_error = "gtx_graph_build: NULL argument";
      return GTX_ERR_ARG;
    }
    // the records have to be what the constructor hands to Graph::add_genomic_region: sorted by position
    // (constructor.cpp:1749-1757) and -- those the region filter below keeps (graph.cpp:60-79: records in front of the
    // region are erased, the list is cut at the first one at or behind region_end) -- inside the reference that came with
    // them (check_if_var_records_match_reference_genome, constructor.cpp:1736).  A kept record outside it used to slip
    // through and leave nodes of length 0 or lost sites behind; a record behind the region is simply dropped, as the
    // reference does (an SV breakpoint record moved by SVLEN lands there when the SV starts inside and ends outside).
    for (uint32_t r = 0; r < n_records; ++r)
    {
      if (records[r].n_alleles < 1 || !records[r].alleles)
      {
        g_last_error = "gtx_graph_build: record " + std::to_string(r) + " has no alleles";
        return GTX_ERR_ARG;
      }
      if (r > 0 && records[r].pos < records[r - 1].pos)
      {
        g_last_error = "gtx_graph_build: records are not sorted by position (record " + std::to_string(r) + ")";
        return GTX_ERR_ARG;
      }
      long const pos = records[r].pos, ref_len = records[r].alleles[0].len;
      if (pos >= region_begin && pos < region_end && pos + ref_len > region_begin + static_cast<long>(reference_len))
      {
        g_last_error = "gtx_graph_build: record " + std::to_string(r) + " at " + std::to_string(pos) + " leaves the reference sequence";
        return GTX_ERR_ARG;
      }
    }
    std::string const refseq(reference, reference_len);
    auto ref_slice = [&](long a, long b)
    {
      long const L = static_cast<long>(refseq.size());
      a = std::min(std::max(a - region_begin, 0l), L);
      b = std::min(std::max(b - region_begin, 0l), L);
      return b > a ? refseq.substr(a, b - a) : std::string();
    };
    // ---- intake + the filters of graph.cpp:48-80
    std::vector<Site> sites;
    for (uint32_t r = 0; r < n_records; ++r)
    {
      gtx_record const & in = records[r];
      if (in.n_alleles < 1 || !in.alleles)
        return GTX_ERR_ARG;
      Site s;
      s.pos = in.pos;
      s.is_sv = in.is_sv != 0;
      for (uint32_t a = 0; a < in.n_alleles; ++a)
      {
        gtx_allele const & ia = in.alleles[a];
        Allele al;
        al.seq.assign(ia.seq ? ia.seq : "", ia.len);
        al.events.assign(ia.events, ia.events + ia.n_events);
        al.anti.assign(ia.anti_events, ia.anti_events + ia.n_anti_events);
        std::sort(al.events.begin(), al.events.end());
        al.events.erase(std::unique(al.events.begin(), al.events.end()), al.events.end());
        std::sort(al.anti.begin(), al.anti.end());
        al.anti.erase(std::unique(al.anti.begin(), al.anti.end()), al.anti.end());
        s.alleles.push_back(std::move(al));
      }
      if (extend_prefix && !s.is_sv) // genomic_region.cpp:236-256
      {
        size_t at = static_cast<size_t>(static_cast<long>(s.pos) - region_begin) + s.ref().size();
        for (;;)
        {
          if (at >= refseq.size() || refseq[at] == 'N')
            break;
          bool related = false;
          for (size_t i = 0; i < s.alleles.size() && !related; ++i)
            for (size_t j = i + 1; j < s.alleles.size() && !related; ++j)
              related = prefix_related(s.alleles[i].seq, s.alleles[j].seq);
          if (!related)
            break;
          for (auto & a : s.alleles)
            a.seq.push_back(refseq[at]);
          ++at;
        }
      }
      // alts with an N or without sequence are ignored (graph.cpp:48-59; after the constructor's prefix extension)
      s.alleles.erase(std::remove_if(s.alleles.begin() + 1, s.alleles.end(),
                                     [](Allele const & a) { return a.seq.empty() || a.seq.find('N') != std::string::npos; }),
                      s.alleles.end());
      if (s.ref().find('N') != std::string::npos || s.ref().find('*') != std::string::npos || s.n_alts() == 0 ||
          static_cast<long>(s.pos) < region_begin)
        continue;
      if (static_cast<long>(s.pos) >= region_end)
        break;
      sites.push_back(std::move(s));
    }
    // ---- merging (graph.cpp:81-240)
    long const n = static_cast<long>(sites.size());
    if (add_all_variants)
    {
      for (long i = 0; i < n; ++i)
        while (i + 1 < n)
        {
          Site & cur = sites[i];
          Site & nxt = sites[i + 1];
          long const gap = static_cast<long>(nxt.pos) - cur.end();
          if (gap > 10)
            break;
          if ((!cur.snps_only() || !nxt.snps_only()) && gap > 2)
            break;
          if (gap >= 0 && (cur.n_alts() > 42 || nxt.n_alts() > 42 || cur.any_longer_than(20) || nxt.any_longer_than(20)))
            break;
          if ((cur.n_alts() + 1) * (nxt.n_alts() + 1) >= MAX_HAPS - 1)
            merge_one_path(nxt, cur);
          else
          {
            if (gap > 0)
            {
              std::string const filler = ref_slice(cur.end(), nxt.pos);
              for (auto & a : cur.alleles)
                a.seq += filler;
            }
            merge_adjacent(nxt, cur);
          }
          if (nxt.n_alts() >= MAX_HAPS - 1)
            nxt.alleles.resize(MAX_HAPS); // REF + (MAX_HAPS - 1) alts
          cur.dead = true;
          ++i;
        }
    }
    else
    {
      for (long i = 0; i < n; ++i)
        while (i + 1 < n && static_cast<long>(sites[i + 1].pos) < sites[i].end())
        {
          Site & cur = sites[i];
          Site & nxt = sites[i + 1];
          if (is_sv_graph && (cur.is_sv || nxt.is_sv))
          {
            if (cur.is_sv && nxt.is_sv)
              merge_one_path(nxt, cur);
            else if (cur.is_sv)
              nxt = cur; // the small variant that overlaps an SV breakpoint is dropped
          }
          else if (cur.n_alts() > 100 || nxt.pos - cur.pos < 4)
            merge_one_path(nxt, cur);
          else
            merge_overlapping(nxt, cur, 4);
          cur.dead = true;
          ++i;
        }
    }
    // ---- clean up (graph.cpp:243-293) and emit nodes (graph.cpp:295-304, 548-625)
    auto g = std::make_unique<gtx_graph>();
    std::string var_dna;
    std::vector<std::pair<Events, Events>> var_events;
    std::vector<std::string> ref_seqs;
    long start = region_begin;
    for (Site & s : sites)
    {
      if (s.dead)
        continue;
      std::string const ref = s.ref();
      s.alleles.erase(std::remove_if(s.alleles.begin() + 1, s.alleles.end(), [&](Allele const & a) { return a.seq == ref; }),
                      s.alleles.end());
      if (s.n_alts() == 0)
        continue;
      if (s.n_alts() >= MAX_HAPS - 1)
        s.alleles.resize(MAX_HAPS - 1); // REF + (MAX_HAPS - 2) alts
      // common suffix, keeping at least one base of every allele (VarRecord::get_common_suffix)
      size_t cut = 0;
      for (;;)
      {
        bool ok = cut + 1 < s.ref().size();
        for (size_t a = 1; a < s.alleles.size() && ok; ++a)
          ok = cut + 1 < s.alleles[a].seq.size() && s.alleles[a].seq[s.alleles[a].seq.size() - 1 - cut] == s.ref()[s.ref().size() - 1 - cut];
        if (!ok)
          break;
        ++cut;
      }
      if (cut)
        for (auto & a : s.alleles)
          a.seq.resize(a.seq.size() - cut);
      std::sort(s.alleles.begin() + 1, s.alleles.end(), [](Allele const & a, Allele const & b) { return a.seq < b.seq; });
      // reference node in front of the site
      long const limit = static_cast<long>(refseq.size()) + region_begin;
      long const node_end = std::max(start, std::min<long>(s.pos, limit));
      g->ref_order.push_back(static_cast<uint32_t>(start + 1));
      ref_seqs.push_back(ref_slice(start, node_end));
      g->ref_nvar.push_back(static_cast<uint32_t>(s.alleles.size()));
      g->ref_first_var.push_back(static_cast<uint32_t>(g->var_order.size()));
      uint32_t const next_ref = static_cast<uint32_t>(g->ref_order.size());
      for (auto & a : s.alleles)
      {
        g->var_order.push_back(s.pos + 1);
        g->var_len.push_back(static_cast<uint32_t>(a.seq.size()));
        g->var_out_ref.push_back(next_ref);
        var_dna += a.seq;
        var_events.emplace_back(std::move(a.events), std::move(a.anti));
      }
      start = static_cast<long>(s.pos) + static_cast<long>(s.ref().size());
    }
    {
      long const limit = static_cast<long>(refseq.size()) + region_begin;
      g->ref_order.push_back(static_cast<uint32_t>(start + 1));
      ref_seqs.push_back(ref_slice(start, std::max(start, limit)));
      g->ref_nvar.push_back(0);
      g->ref_first_var.push_back(GTX_INVALID_ID);
    }
    for (auto const & s : ref_seqs)
    {
      g->ref_dna_off.push_back(static_cast<uint32_t>(g->dna.size()));
      g->ref_len.push_back(static_cast<uint32_t>(s.size()));
      g->dna += s;
    }
    {
      size_t off = g->dna.size(), p = 0;
      g->dna += var_dna;
      for (uint32_t len : g->var_len)
      {
        g->var_dna_off.push_back(static_cast<uint32_t>(off + p));
        p += len;
      }
    }
    g->event_off.push_back(0);
    for (auto const & ev : var_events)
    {
      g->event_val.insert(g->event_val.end(), ev.first.begin(), ev.first.end());
      g->event_off.push_back(static_cast<uint32_t>(g->event_val.size()));
      g->event_val.insert(g->event_val.end(), ev.second.begin(), ev.second.end());
      g->event_off.push_back(static_cast<uint32_t>(g->event_val.size()));
    }
    *out = g.release();
    return GTX_OK;
  }

  int gtx_graph_get_view(const gtx_graph * g, gtx_graph_view * out)
  {
    if (!g || !out)
      return GTX_ERR_ARG;
    out->n_ref = static_cast<uint32_t>(g->ref_order.size());
    out->n_var = static_cast<uint32_t>(g->var_order.size());
    out->ref_order = g->ref_order.data();
    out->ref_len = g->ref_len.data();
    out->ref_dna_off = g->ref_dna_off.data();
    out->ref_nvar = g->ref_nvar.data();
    out->ref_first_var = g->ref_first_var.data();
    out->var_order = g->var_order.data();
    out->var_len = g->var_len.data();
    out->var_dna_off = g->var_dna_off.data();
    out->var_out_ref = g->var_out_ref.data();
    out->dna = g->dna.data();
    out->dna_len = g->dna.size();
    out->event_off = g->event_val.empty() ? nullptr : g->event_off.data();
    out->event_val = g->event_val.empty() ? nullptr : g->event_val.data();
    return GTX_OK;
  }

  int gtx_graph_sv_table(const gtx_graph * g, char * out, uint64_t cap, uint64_t * len)
  {
    if (!g || !len)
      return GTX_ERR_ARG;
    *len = g->sv_table.size();
    if (out && cap)
      std::memcpy(out, g->sv_table.data(), static_cast<size_t>(std::min<uint64_t>(cap, g->sv_table.size())));
    return GTX_OK;
  }

  void gtx_graph_destroy(gtx_graph * g) { delete g; }
}
