// gto_capi.cpp -- C entry points of the CPU ORACLE (test infrastructure, not product code).
// Loaded with ctypes from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
#include "gto.hpp"
#include "gto_vcf.hpp"
#include "gto_sv.hpp"
#include "gto_discovery.hpp"
#include "gto_shrink.hpp"

#include <cctype>
#include <memory>
#include <sstream>

using namespace gto;

namespace
{
thread_local std::string g_error;

struct Handle
{
  Params par;
  Graph graph;
  PHIndex index;
};

struct GenoHandle
{
  Handle * h;
  std::unique_ptr<Genotyper> g;
};

// records text: one record per line, fields separated by blanks:  pos0  REF  ALT1,ALT2  [INFO]
// INFO carries GT_ID / GT_ANTI_HAPLOTYPE exactly as src/graph/constructor.cpp:1540-1588 reads them
// (only for records with a single alt), and -- for tests that set events directly like test/graph/test_graph.cpp --
// RE= / RA= (reference allele events / anti events) and E<i>= / A<i>= (events / anti events of alt i), comma separated.
std::vector<VarRecord> parse_records(std::string const & text)
{
  std::vector<VarRecord> out;
  std::istringstream in(text);
  std::string line;
  while (std::getline(in, line))
  {
    if (line.empty() || line[0] == '#')
      continue;
    std::istringstream ls(line);
    VarRecord r;
    std::string alts, info;
    long pos;
    ls >> pos >> r.ref >> alts;
    ls >> info;
    r.pos = static_cast<uint32_t>(pos);
    std::size_t a = 0;
    while (a <= alts.size())
    {
      std::size_t const b = alts.find(',', a);
      AltAllele alt;
      alt.seq = alts.substr(a, b == std::string::npos ? std::string::npos : b - a);
      r.alts.push_back(alt);
      if (b == std::string::npos)
        break;
      a = b + 1;
    }
    auto split_longs = [](std::string const & val)
    {
      std::vector<long> out;
      std::size_t c = 0;
      while (c <= val.size())
      {
        std::size_t const d = val.find(',', c);
        std::string const tok = val.substr(c, d == std::string::npos ? std::string::npos : d - c);
        if (!tok.empty())
          out.push_back(std::stol(tok));
        if (d == std::string::npos)
          break;
        c = d + 1;
      }
      return out;
    };
    if (!info.empty() && info != ".")
    {
      std::size_t s = 0;
      while (s <= info.size())
      {
        std::size_t const e = info.find(';', s);
        std::string const kv = info.substr(s, e == std::string::npos ? std::string::npos : e - s);
        std::size_t const eq = kv.find('=');
        if (eq != std::string::npos)
        {
          std::string const key = kv.substr(0, eq), val = kv.substr(eq + 1);
          if (key == "SV") // the record is a structural variant whose alleles the caller synthesised (VarRecord::is_sv)
            r.is_sv = val != "0";
          else if (key == "RE")
            for (long x : split_longs(val))
              r.ref_events.insert(x);
          else if (key == "RA")
            for (long x : split_longs(val))
              r.ref_anti_events.insert(x);
          else if ((key[0] == 'E' || key[0] == 'A') && key.size() > 1 && std::isdigit(static_cast<unsigned char>(key[1])))
          {
            std::size_t const ai = std::stoul(key.substr(1));
            if (ai < r.alts.size())
              for (long x : split_longs(val))
                (key[0] == 'E' ? r.alts[ai].events : r.alts[ai].anti_events).insert(x);
          }
        }
        if (e == std::string::npos)
          break;
        s = e + 1;
      }
    }
    if (r.alts.size() == 1 && !info.empty() && info != ".")
    {
      std::size_t s = 0;
      while (s <= info.size())
      {
        std::size_t const e = info.find(';', s);
        std::string const kv = info.substr(s, e == std::string::npos ? std::string::npos : e - s);
        std::size_t const eq = kv.find('=');
        if (eq != std::string::npos)
        {
          std::string const key = kv.substr(0, eq), val = kv.substr(eq + 1);
          if (key == "GT_ID")
          {
            long const id = std::stol(val);
            r.ref_events.insert(-id);
            r.alts[0].events.insert(id);
          }
          else if (key == "GT_ANTI_HAPLOTYPE")
          {
            std::size_t c = 0;
            while (c <= val.size())
            {
              std::size_t const d = val.find(',', c);
              r.alts[0].anti_events.insert(std::stol(val.substr(c, d == std::string::npos ? std::string::npos : d - c)));
              if (d == std::string::npos)
                break;
              c = d + 1;
            }
          }
        }
        if (e == std::string::npos)
          break;
        s = e + 1;
      }
    }
    out.push_back(std::move(r));
  }
  return out;
}

void put_path_stream(std::vector<uint32_t> & out, GenotypePaths const & g)
{
  out.push_back(static_cast<uint32_t>(g.paths.size()));
  out.push_back(g.longest_path_length);
  for (auto const & p : g.paths)
  {
    out.push_back(p.start);
    out.push_back(p.end);
    out.push_back(p.read_start_index);
    out.push_back(p.read_end_index);
    out.push_back(p.mismatches);
    out.push_back(static_cast<uint32_t>(p.var_order.size()));
    for (std::size_t i = 0; i < p.var_order.size(); ++i)
    {
      out.push_back(p.var_order[i]);
      out.push_back(static_cast<uint32_t>(p.nums[i].size()));
      for (uint16_t n : p.nums[i])
        out.push_back(n);
    }
  }
}

ReadRecord make_record(long i, uint8_t const * codes, uint32_t const * offs, uint16_t const * flags, int32_t const * tid,
                       int32_t const * mtid, int64_t const * pos, int64_t const * isize, uint8_t const * mapq,
                       uint8_t const * score_diff, uint64_t const * name, int32_t const * sample, int32_t const * rg)
{
  ReadRecord r;
  r.seq.assign(codes + offs[i], codes + offs[i + 1]);
  for (auto & c : r.seq)
    if ((c & 15) == 0)
      c = 15; // '=' becomes N when the character is assigned to a seqan Iupac
  r.flag = flags ? flags[i] : 0;
  r.tid = tid ? tid[i] : 0;
  r.mtid = mtid ? mtid[i] : 0;
  r.pos = pos ? pos[i] : 0;
  r.isize = isize ? isize[i] : 0;
  r.mapq = mapq ? mapq[i] : 60;
  r.score_diff = score_diff ? score_diff[i] : 0;
  r.name = name ? std::to_string(name[i]) : std::to_string(i);
  r.sample = sample ? sample[i] : 0;
  r.rg = rg ? rg[i] : 0;
  return r;
}

} // namespace

extern "C"
{
  char const * gto_last_error() { return g_error.c_str(); }

  void * gto_new(char const * reference, long region_begin, char const * records_text, int is_sv_graph, int hq_reads,
                 int force_both, long max_index_labels, int add_all_variants, int extend_prefix)
  {
    try
    {
      auto h = std::make_unique<Handle>();
      h->par.is_sv_graph = is_sv_graph != 0;
      h->par.hq_reads = hq_reads != 0;
      h->par.force_align_both_orientations = force_both != 0;
      h->par.max_index_labels = max_index_labels;
      h->graph.is_sv_graph = is_sv_graph != 0;
      h->graph.region_begin = region_begin;
      // GenomicRegion::end: the reference sequence handed over IS the region (read_reference_genome, constructor.cpp:1614-1616)
      h->graph.region_end = region_begin + static_cast<long>(std::strlen(reference));
      h->graph.add_all_variants = add_all_variants != 0;
      std::vector<VarRecord> records = parse_records(records_text);
      if (extend_prefix)
        for (auto & r : records)
          r.extend_while_prefix_related(reference, region_begin);
      h->graph.add_genomic_region(reference, std::move(records));
      h->graph.create_special_positions();
      h->index = index_graph(h->graph, max_index_labels);
      return h.release();
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return nullptr;
    }
  }

  void gto_free(void * p) { delete static_cast<Handle *>(p); }

  // out: n_ref, n_var, n_special, dna bytes (ref nodes then var nodes), n_events (events + anti events over all var nodes)
  void gto_graph_counts(void * p, long * out)
  {
    Graph const & g = static_cast<Handle *>(p)->graph;
    long dna = 0, ev = 0;
    for (auto const & r : g.ref_nodes)
      dna += static_cast<long>(r.label.dna.size());
    for (auto const & v : g.var_nodes)
    {
      dna += static_cast<long>(v.label.dna.size());
      ev += static_cast<long>(v.events.size() + v.anti_events.size());
    }
    out[0] = static_cast<long>(g.ref_nodes.size());
    out[1] = static_cast<long>(g.var_nodes.size());
    out[2] = static_cast<long>(g.ref_reach_poses.size());
    out[3] = dna;
    out[4] = ev;
  }

  // events: per var node  n_events, n_anti, then the values (sorted ascending) as int64
  void gto_graph_dump(void * p, uint32_t * ref_order, uint32_t * ref_len, uint32_t * ref_nvar, uint32_t * ref_first_var,
                      uint32_t * var_order, uint32_t * var_len, uint32_t * var_out_ref, char * dna, uint32_t * ref_reach_poses,
                      uint32_t * actual_poses, int64_t * events)
  {
    Graph const & g = static_cast<Handle *>(p)->graph;
    long d = 0;
    for (std::size_t r = 0; r < g.ref_nodes.size(); ++r)
    {
      auto const & n = g.ref_nodes[r];
      ref_order[r] = n.label.order;
      ref_len[r] = static_cast<uint32_t>(n.label.dna.size());
      ref_nvar[r] = static_cast<uint32_t>(n.out_var_ids.size());
      ref_first_var[r] = n.out_var_ids.empty() ? INVALID_ID : n.out_var_ids[0];
      std::memcpy(dna + d, n.label.dna.data(), n.label.dna.size());
      d += static_cast<long>(n.label.dna.size());
    }
    long e = 0;
    for (std::size_t v = 0; v < g.var_nodes.size(); ++v)
    {
      auto const & n = g.var_nodes[v];
      var_order[v] = n.label.order;
      var_len[v] = static_cast<uint32_t>(n.label.dna.size());
      var_out_ref[v] = n.out_ref_id;
      std::memcpy(dna + d, n.label.dna.data(), n.label.dna.size());
      d += static_cast<long>(n.label.dna.size());
      std::set<long> ev(n.events.begin(), n.events.end()), an(n.anti_events.begin(), n.anti_events.end());
      events[e++] = static_cast<int64_t>(ev.size());
      events[e++] = static_cast<int64_t>(an.size());
      for (long x : ev)
        events[e++] = x;
      for (long x : an)
        events[e++] = x;
    }
    for (std::size_t i = 0; i < g.ref_reach_poses.size(); ++i)
    {
      ref_reach_poses[i] = g.ref_reach_poses[i];
      actual_poses[i] = g.actual_poses[i];
    }
  }

  long gto_all_ref(void * p, char * out, long cap)
  {
    std::string const s = static_cast<Handle *>(p)->graph.get_all_ref();
    if (static_cast<long>(s.size()) <= cap)
      std::memcpy(out, s.data(), s.size());
    return static_cast<long>(s.size());
  }

  long gto_index_num_keys(void * p) { return static_cast<long>(static_cast<Handle *>(p)->index.hamming0.size()); }

  long gto_index_num_labels(void * p)
  {
    long n = 0;
    for (auto const & kv : static_cast<Handle *>(p)->index.hamming0)
      n += static_cast<long>(kv.second.size());
    return n;
  }

  // keys ascending; counts per key; labels (start,end,var) in bucket order
  void gto_index_dump(void * p, uint64_t * keys, uint32_t * counts, uint32_t * labels)
  {
    auto const & m = static_cast<Handle *>(p)->index.hamming0;
    std::vector<uint64_t> ks;
    ks.reserve(m.size());
    for (auto const & kv : m)
      ks.push_back(kv.first);
    std::sort(ks.begin(), ks.end());
    long l = 0;
    for (std::size_t i = 0; i < ks.size(); ++i)
    {
      auto const & v = m.at(ks[i]);
      keys[i] = ks[i];
      counts[i] = static_cast<uint32_t>(v.size());
      for (auto const & lab : v)
      {
        labels[l++] = lab.start_index;
        labels[l++] = lab.end_index;
        labels[l++] = lab.variant_id;
      }
    }
  }

  long gto_index_get(void * p, uint64_t key, uint32_t * out, long cap)
  {
    auto const v = static_cast<Handle *>(p)->index.get(key);
    for (long i = 0; i < static_cast<long>(v.size()) && i < cap; ++i)
    {
      out[3 * i] = v[i].start_index;
      out[3 * i + 1] = v[i].end_index;
      out[3 * i + 2] = v[i].variant_id;
    }
    return static_cast<long>(v.size());
  }

  int gto_index_check(void * p)
  {
    auto * h = static_cast<Handle *>(p);
    return h->index.check(h->graph) ? 1 : 0;
  }

  // per k-mer label lists of one read: exact lists then Hamming-1 lists.  stream: n_k, then for each of the 2*n_k lists:
  // count, (start,end,var)*count
  long gto_query_read(void * p, uint8_t const * codes, long len, uint32_t * out, long cap)
  {
    auto * h = static_cast<Handle *>(p);
    std::vector<uint8_t> read(codes, codes + len);
    auto const r0 = query_index(read, h->index);
    auto const r1 = query_index_hamming1(read, h->index);
    std::vector<uint32_t> s;
    s.push_back(static_cast<uint32_t>(r0.size()));
    for (auto const * lists : {&r0, &r1})
      for (auto const & l : *lists)
      {
        s.push_back(static_cast<uint32_t>(l.size()));
        for (auto const & x : l)
        {
          s.push_back(x.start_index);
          s.push_back(x.end_index);
          s.push_back(x.variant_id);
        }
      }
    if (static_cast<long>(s.size()) <= cap)
      std::memcpy(out, s.data(), s.size() * 4);
    return static_cast<long>(s.size());
  }

  // align_read() for n records; stream per read: forward GenotypePaths then reverse GenotypePaths (see put_path_stream)
  long gto_align(void * p, long n, uint8_t const * codes, uint32_t const * offs, uint16_t const * flags, int32_t const * tid,
                 int32_t const * mtid, int64_t const * isize, uint32_t * out, long cap)
  {
    try
    {
      auto * h = static_cast<Handle *>(p);
      std::vector<uint32_t> s;
      for (long i = 0; i < n; ++i)
      {
        ReadRecord const r = make_record(i, codes, offs, flags, tid, mtid, nullptr, isize, nullptr, nullptr, nullptr, nullptr, nullptr);
        auto const gp = align_read(r, h->index, h->graph, h->par);
        put_path_stream(s, gp.first);
        put_path_stream(s, gp.second);
      }
      if (static_cast<long>(s.size()) <= cap)
        std::memcpy(out, s.data(), s.size() * 4);
      return static_cast<long>(s.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  void * gto_genotyper_new(void * p, long n_samples, long n_rg)
  {
    auto * h = static_cast<Handle *>(p);
    auto * g = new GenoHandle;
    g->h = h;
    g->g = std::make_unique<Genotyper>(h->graph, h->index, h->par, static_cast<std::size_t>(n_samples), static_cast<std::size_t>(n_rg));
    return g;
  }

  void gto_genotyper_free(void * p) { delete static_cast<GenoHandle *>(p); }

  int gto_genotyper_push(void * p, long n, uint8_t const * codes, uint32_t const * offs, uint16_t const * flags,
                         int32_t const * tid, int32_t const * mtid, int64_t const * pos, int64_t const * isize,
                         uint8_t const * mapq, uint8_t const * score_diff, uint64_t const * name, int32_t const * sample,
                         int32_t const * rg)
  {
    try
    {
      auto * g = static_cast<GenoHandle *>(p);
      for (long i = 0; i < n; ++i)
        g->g->push(make_record(i, codes, offs, flags, tid, mtid, pos, isize, mapq, score_diff, name, sample, rg));
      return 0;
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // same as gto_genotyper_push plus the fields of the SV-mode record filter (any of them may be NULL = 0)
  int gto_genotyper_push_ex(void * p, long n, uint8_t const * codes, uint32_t const * offs, uint16_t const * flags,
                            int32_t const * tid, int32_t const * mtid, int64_t const * pos, int64_t const * isize,
                            uint8_t const * mapq, uint8_t const * score_diff, uint64_t const * name, int32_t const * sample,
                            int32_t const * rg, int64_t const * mpos, uint32_t const * n_cigar, uint32_t const * cigar_front,
                            uint32_t const * cigar_back)
  {
    try
    {
      auto * g = static_cast<GenoHandle *>(p);
      for (long i = 0; i < n; ++i)
      {
        ReadRecord r = make_record(i, codes, offs, flags, tid, mtid, pos, isize, mapq, score_diff, name, sample, rg);
        r.mpos = mpos ? mpos[i] : 0;
        r.n_cigar = n_cigar ? n_cigar[i] : 0;
        r.cigar_front = cigar_front ? cigar_front[i] : 0;
        r.cigar_back = cigar_back ? cigar_back[i] : 0;
        g->g->push(r);
      }
      return 0;
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  void gto_genotyper_set_coverage(void * p, double const * avg_cov_by_readlen, long n, int no_filter_on_coverage)
  {
    auto & g = *static_cast<GenoHandle *>(p)->g;
    g.avg_cov_by_readlen.assign(avg_cov_by_readlen, avg_cov_by_readlen + n);
    g.no_filter_on_coverage = no_filter_on_coverage != 0;
  }

  // end of the record stream (SV calling scores the reads whose mate never came)
  void gto_genotyper_finish(void * p) { static_cast<GenoHandle *>(p)->g->finish(); }

  long gto_genotyper_num_haplotypes(void * p) { return static_cast<long>(static_cast<GenoHandle *>(p)->g->writer.haplotypes.size()); }

  // Canonical score stream (u32 words), per haplotype:
  //   id, num, clipped_reads, mapq_squared lo, hi,
  //   per allele: clipped_bp lo,hi, mapq_squared lo,hi, score_diff, mismatches, r1f, r1r, r2f, r2r
  //   per sample: max_log_score, ambiguous_depth, ambiguous_depth_alt, alt_proper_pair_depth,
  //               gt_coverage[num], log_score[num(num+1)/2],
  //               per allele: n_conn, then n_conn * (hap2, counts[num(hap2)])   (hap2 ascending)
  long gto_scores_dump(void * p, uint32_t * out, long cap)
  {
    auto const & W = static_cast<GenoHandle *>(p)->g->writer;
    std::vector<uint32_t> s;
    auto put64 = [&s](uint64_t x)
    {
      s.push_back(static_cast<uint32_t>(x));
      s.push_back(static_cast<uint32_t>(x >> 32));
    };
    for (auto const & h : W.haplotypes)
    {
      s.push_back(h.id);
      s.push_back(h.num);
      s.push_back(h.clipped_reads);
      put64(h.mapq_squared);
      for (uint16_t a = 0; a < h.num; ++a)
      {
        put64(h.per_allele[a].clipped_bp);
        put64(h.per_allele[a].mapq_squared);
        s.push_back(h.per_allele[a].score_diff);
        s.push_back(h.per_allele[a].mismatches);
        s.push_back(h.read_strand[a].r1_forward);
        s.push_back(h.read_strand[a].r1_reverse);
        s.push_back(h.read_strand[a].r2_forward);
        s.push_back(h.read_strand[a].r2_reverse);
      }
      for (auto const & hs : h.hap_samples)
      {
        s.push_back(hs.max_log_score);
        s.push_back(hs.ambiguous_depth);
        s.push_back(hs.ambiguous_depth_alt);
        s.push_back(hs.alt_proper_pair_depth);
        for (uint16_t c : hs.gt_coverage)
          s.push_back(c);
        for (uint16_t c : hs.log_score)
          s.push_back(c);
        for (auto const & m : hs.connections)
        {
          s.push_back(static_cast<uint32_t>(m.size()));
          for (auto const & kv : m)
          {
            s.push_back(kv.first);
            for (uint16_t c : kv.second)
              s.push_back(c);
          }
        }
      }
    }
    if (static_cast<long>(s.size()) <= cap)
      std::memcpy(out, s.data(), s.size() * 4);
    return static_cast<long>(s.size());
  }

  // sample calls (Vcf::add_haplotype): u32 words, per haplotype, per sample:
  //   gt_first, gt_second, gq, ref_total_depth, alt_total_depth, ambiguous_depth, alt_proper_pair_depth, n_phred, phred...
  long gto_calls_dump(void * p, uint32_t * out, long cap)
  {
    auto const calls = static_cast<GenoHandle *>(p)->g->sample_calls();
    std::vector<uint32_t> s;
    for (auto const & row : calls)
      for (auto const & c : row)
      {
        s.insert(s.end(), {c.gt_first, c.gt_second, c.gq, c.ref_total_depth, c.alt_total_depth, c.ambiguous_depth, c.alt_proper_pair_depth,
                           static_cast<uint32_t>(c.phred.size())});
        s.insert(s.end(), c.phred.begin(), c.phred.end());
      }
    if (static_cast<long>(s.size()) <= cap)
      std::memcpy(out, s.data(), s.size() * 4);
    return static_cast<long>(s.size());
  }

  // phase flags (hts_parallel_reader.cpp:782-904): rows of (hap1, allele1, hap2, allele2, flags); an outer key without
  // any flag under it is one row with hap2 = allele2 = 0xFFFF, flags = 0.  Returns the number of rows.
  long gto_phase_flags(void * p, int32_t * out, long cap_rows)
  {
    auto ph = static_cast<GenoHandle *>(p)->g->phase_flags();
    long n = 0;
    auto put = [&](int a, int b, int c, int d, int f)
    {
      if (n < cap_rows)
      {
        int32_t * o = out + 5 * n;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = f;
      }
      ++n;
    };
    for (auto const & r : ph)
    {
      if (r.second.empty())
        put(r.first.first, r.first.second, 0xFFFF, 0xFFFF, 0);
      for (auto const & e : r.second)
        put(r.first.first, r.first.second, e.first.first, e.first.second, e.second);
    }
    return n;
  }

  // ReferenceDepth::depths of one sample (SV calling): returns the number of positions; out may be NULL
  long gto_reference_depth(void * p, long sample, uint16_t * out, long cap)
  {
    auto * g = static_cast<GenoHandle *>(p);
    auto const & d = g->g->reference_depth.depths;
    if (sample < 0 || sample >= static_cast<long>(d.size()))
      return 0;
    long const n = static_cast<long>(d[sample].size());
    for (long i = 0; out && i < n && i < cap; ++i)
      out[i] = d[sample][i];
    return n;
  }

  // VCF records of the genotyper's sites (gto_vcf.hpp).  sample_names: '\n'-separated.  Returns the text length; copies
  // min(length, cap) bytes.
  long gto_vcf_records(void * p, char const * contig, char const * sample_names, uint32_t region_begin, uint32_t region_end,
                       int filter_zero_qual, char const * suffix_id, char * out, long cap)
  {
    vcf::WriteOptions o;
    o.contig = contig;
    {
      std::stringstream ss(sample_names ? sample_names : "");
      std::string n;
      while (std::getline(ss, n, '\n'))
        if (!n.empty())
          o.sample_names.push_back(n);
    }
    o.region_begin = region_begin;
    o.region_end = region_end;
    o.filter_zero_qual = filter_zero_qual != 0;
    o.variant_suffix_id = suffix_id ? suffix_id : "";
    std::string const text = vcf::records(*static_cast<GenoHandle *>(p)->g, o);
    if (out && cap > 0)
      std::memcpy(out, text.data(), static_cast<std::size_t>(std::min<long>(cap, static_cast<long>(text.size()))));
    return static_cast<long>(text.size());
  }

  // the sites-only records of vcf_merge_and_filter with the genotyper's own phasing flags (gto_vcf.hpp: sites)
  long gto_vcf_sites(void * p, char const * contig, char * out, long cap)
  {
    Genotyper const & g = *static_cast<GenoHandle *>(p)->g;
    std::string const text = vcf::sites(g, contig, g.phase_flags());
    if (out && cap > 0)
      std::memcpy(out, text.data(), static_cast<std::size_t>(std::min<long>(cap, static_cast<long>(text.size()))));
    return static_cast<long>(text.size());
  }

  // VCF records of an SV graph's calls (gto_sv.hpp: reformat_sv_vcf_records and the merge of genotype_sv).  sv_table: the text
  // form of Graph::SVs; reference / first_pos: the region's reference sequence and the 1-based position of its first base.
  // Returns the text length, or -1 (gto_last_error).
  long gto_vcf_records_sv(void * p, char const * contig, char const * sample_names, uint32_t region_begin, uint32_t region_end,
                          char const * sv_table, char const * reference, uint32_t first_pos, char * out, long cap)
  {
    try
    {
      vcf::WriteOptions o;
      o.contig = contig;
      std::stringstream ss(sample_names ? sample_names : "");
      std::string n;
      while (std::getline(ss, n, '\n'))
        if (!n.empty())
          o.sample_names.push_back(n);
      o.region_begin = region_begin;
      o.region_end = region_end;
      vcf::RegionReference rr;
      rr.reference = reference ? reference : "";
      rr.first_pos = first_pos;
      std::string const text = vcf::records_sv(*static_cast<GenoHandle *>(p)->g, o, vcf::parse_sv_table(sv_table ? sv_table : ""), rr);
      if (out && cap > 0)
        std::memcpy(out, text.data(), static_cast<std::size_t>(std::min<long>(cap, static_cast<long>(text.size()))));
      return static_cast<long>(text.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // the final VCF of a small-variant graph (gto_sv.hpp: records_final -- vcf_merge_and_break with the break-down).  reference /
  // first_pos: the region's reference sequence and the 1-based position of its first base.  Returns the text length, or -1.
  long gto_vcf_records_final(void * p, char const * contig, char const * sample_names, uint32_t region_begin, uint32_t region_end,
                             int filter_zero_qual, char const * reference, uint32_t first_pos, int no_variant_overlapping, char * out, long cap)
  {
    try
    {
      vcf::WriteOptions o;
      o.contig = contig;
      std::stringstream ss(sample_names ? sample_names : "");
      std::string n;
      while (std::getline(ss, n, '\n'))
        if (!n.empty())
          o.sample_names.push_back(n);
      o.region_begin = region_begin;
      o.region_end = region_end;
      o.filter_zero_qual = filter_zero_qual != 0;
      vcf::RegionReference rr;
      rr.reference = reference ? reference : "";
      rr.first_pos = first_pos;
      std::string const text = vcf::records_final(*static_cast<GenoHandle *>(p)->g, o, rr, no_variant_overlapping != 0);
      if (out && cap > 0)
        std::memcpy(out, text.data(), static_cast<std::size_t>(std::min<long>(cap, static_cast<long>(text.size()))));
      return static_cast<long>(text.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // make_call_based_on_coverage of one SV over a depth track given directly (unit tests): out = coverage[0], coverage[1], PL x 3
  int gto_coverage_call(char const * sv_table, long sv_index, uint16_t const * depth, long n_depth, uint32_t reference_offset, uint32_t * out)
  {
    try
    {
      auto const svs = vcf::parse_sv_table(sv_table);
      ReferenceDepth rd;
      rd.reference_offset = reference_offset;
      rd.depths.assign(1, std::vector<uint16_t>(depth, depth + n_depth));
      vcf::SampleCall const c = vcf::make_call_based_on_coverage(0, svs.at(static_cast<std::size_t>(sv_index)), rd);
      out[0] = c.coverage[0];
      out[1] = c.coverage[1];
      out[2] = c.phred[0];
      out[3] = c.phred[1];
      out[4] = c.phred[2];
      return 0;
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return 1;
    }
  }

  uint16_t gto_binned_pl(unsigned pl) { return vcf::binned_pl(pl); }
  double gto_p_hwe_excess_het(int het, int hom1, int hom2) { return vcf::p_hwe_excess_het(het, hom1, hom2); }

  // dst += src (Genotyper::merge_from: test infrastructure for sharding the oracle over host threads); 0 = ok
  int gto_genotyper_merge(void * dst, void * src)
  {
    try
    {
      static_cast<GenoHandle *>(dst)->g->merge_from(*static_cast<GenoHandle *>(src)->g);
      return 0;
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return 1;
    }
  }

  void gto_genotyper_counts(void * p, long * out)
  {
    auto * g = static_cast<GenoHandle *>(p);
    out[0] = g->g->num_records;
    out[1] = g->g->num_duplicated;
    long parked = 0;
    for (auto const & m : g->g->maps)
      parked += static_cast<long>(m.size());
    out[2] = parked;
  }

  // ---- discovery, first pass (gto_discovery.hpp): reads as arrays -> the canonical word stream of the surviving events
  // codes: 4-bit BAM codes of all reads back to back (code_off[n + 1]), qual the same shape, cigar: raw BAM words (cigar_off[n + 1])
  long gto_first_pass(char const * reference, long region_begin, long bucket_size, long n, int32_t const * pos, uint16_t const * flag,
                      uint8_t const * mapq, uint32_t const * cigar, uint32_t const * cigar_off, uint8_t const * codes, uint8_t const * qual,
                      uint32_t const * code_off, uint32_t * out, long cap)
  {
    try
    {
      static char const NT16[] = "=ACMGRSVTWYHKDBN";
      std::vector<gto::disc::Read> reads(static_cast<std::size_t>(n));
      for (long i = 0; i < n; ++i)
      {
        gto::disc::Read & r = reads[i];
        r.pos = pos[i];
        r.flag = flag[i];
        r.mapq = mapq[i];
        r.cigar.assign(cigar + cigar_off[i], cigar + cigar_off[i + 1]);
        for (uint32_t k = code_off[i]; k < code_off[i + 1]; ++k)
          r.sequence.push_back(NT16[codes[k] & 15]);
        r.qual.assign(qual + code_off[i], qual + code_off[i + 1]);
      }
      gto::disc::FirstPass fp;
      fp.run(reads, std::string(reference), region_begin, bucket_size);
      std::vector<uint32_t> const s = fp.dump();
      if (static_cast<long>(s.size()) <= cap)
        std::memcpy(out, s.data(), s.size() * 4);
      return static_cast<long>(s.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // the same pass to its end (the sample's haplotype map, the indels that are left) as the word stream of FirstPass::dump(Result);
  // ev_out / ro_out (may be NULL): the per-read events and read states in the layout of gtx_disc_event / gtx_disc_read_out, for
  // tests that feed the product's host stage without a device (ev_cap events; returns -2 when there are more)
  long gto_first_pass_full(char const * reference, long region_begin, long bucket_size, long file_i, long n, int32_t const * pos, uint16_t const * flag,
                           uint8_t const * mapq, uint32_t const * cigar, uint32_t const * cigar_off, uint8_t const * codes, uint8_t const * qual,
                           uint32_t const * code_off, uint32_t * out, long cap, void * ev_out, long ev_cap, long * n_events, void * ro_out)
  {
    try
    {
      static char const NT16[] = "=ACMGRSVTWYHKDBN";
      std::vector<gto::disc::Read> reads(static_cast<std::size_t>(n));
      for (long i = 0; i < n; ++i)
      {
        gto::disc::Read & r = reads[i];
        r.pos = pos[i];
        r.flag = flag[i];
        r.mapq = mapq[i];
        r.cigar.assign(cigar + cigar_off[i], cigar + cigar_off[i + 1]);
        for (uint32_t k = code_off[i]; k < code_off[i + 1]; ++k)
          r.sequence.push_back(NT16[codes[k] & 15]);
        r.qual.assign(qual + code_off[i], qual + code_off[i + 1]);
      }
      gto::disc::FirstPass fp;
      fp.file_i = file_i;
      fp.run(reads, std::string(reference), region_begin, bucket_size);
      static_assert(sizeof(gto::disc::FirstPass::RawEvent) == 20 && sizeof(gto::disc::FirstPass::ReadOut) == 16, "layout of gtx_disc_event / gtx_disc_read_out");
      if (n_events)
        *n_events = static_cast<long>(fp.raw_events.size());
      if (ev_out)
      {
        if (static_cast<long>(fp.raw_events.size()) > ev_cap)
          return -2;
        if (!fp.raw_events.empty())
          std::memcpy(ev_out, fp.raw_events.data(), fp.raw_events.size() * sizeof(fp.raw_events[0]));
      }
      if (ro_out && n)
        std::memcpy(ro_out, fp.read_outs.data(), fp.read_outs.size() * sizeof(fp.read_outs[0]));
      fp.run_haplotypes(std::string(reference), region_begin, bucket_size);
      std::vector<uint32_t> const s = gto::disc::FirstPass::dump(fp.result());
      if (static_cast<long>(s.size()) <= cap)
        std::memcpy(out, s.data(), s.size() * 4);
      return static_cast<long>(s.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // merge_haplotypes2 + the union of the indels over two such word streams (the first may be empty: nothing merged yet)
  long gto_disc_merge(uint32_t const * into, long n_into, uint32_t const * from, long n_from, uint32_t * out, long cap)
  {
    try
    {
      using FP = gto::disc::FirstPass;
      auto parse = [](uint32_t const * w, long n)
      {
        FP::Result r;
        long at = 0;
        auto need = [&](long k) { if (at + k > n) throw std::runtime_error("gto_disc_merge: truncated stream"); };
        auto event = [&]()
        {
          need(3);
          gto::disc::Event e;
          e.pos = w[at];
          e.type = static_cast<char>(w[at + 1]);
          long const len = w[at + 2];
          at += 3;
          need(len);
          for (long k = 0; k < len; ++k)
            e.sequence.push_back(static_cast<char>(w[at + k]));
          at += len;
          return e;
        };
        if (n == 0)
          return r;
        need(1);
        long const n_indels = w[at++];
        for (long i = 0; i < n_indels; ++i)
        {
          gto::disc::Event e = event();
          gto::disc::EventSupport s;
          need(17);
          s.hq_count = w[at]; s.lq_count = w[at + 1]; s.proper_pairs = w[at + 2]; s.first_in_pairs = w[at + 3]; s.sequence_reversed = w[at + 4];
          s.clipped = w[at + 5]; s.max_mapq = w[at + 6]; s.max_distance = w[at + 7]; s.uniq_pos1 = static_cast<int32_t>(w[at + 8]);
          s.uniq_pos2 = static_cast<int32_t>(w[at + 9]); s.uniq_pos3 = static_cast<int32_t>(w[at + 10]); s.span = w[at + 11];
          s.has_realignment_support = w[at + 12]; s.has_indel_good_support = w[at + 13]; s.max_log_qual = w[at + 14];
          s.max_log_qual_file_i = static_cast<int32_t>(w[at + 15]);
          long const n_phase = w[at + 16];
          at += 17;
          for (long k = 0; k < n_phase; ++k)
          {
            gto::disc::Event pe = event();
            need(1);
            s.phase[pe] = static_cast<uint16_t>(w[at++]);
          }
          r.indels.insert({e, s});
        }
        need(1);
        long const n_haps = w[at++];
        for (long i = 0; i < n_haps; ++i)
        {
          gto::disc::Event e = event();
          FP::Thap t;
          for (std::set<gto::disc::Event> * set : {&t.ever_together, &t.always_together})
          {
            need(1);
            long const m = w[at++];
            for (long k = 0; k < m; ++k)
              set->insert(event());
          }
          r.haplotypes.insert({e, t});
        }
        if (at != n)
          throw std::runtime_error("gto_disc_merge: words behind the stream");
        return r;
      };
      FP::Result a = parse(into, n_into), b = parse(from, n_from);
      FP::merge(a, b);
      std::vector<uint32_t> const s = FP::dump(a);
      if (static_cast<long>(s.size()) <= cap)
        std::memcpy(out, s.data(), s.size() * 4);
      return static_cast<long>(s.size());
    }
    catch (std::exception const & e)
    {
      g_error = e.what();
      return -1;
    }
  }

  // ---- small known-answer helpers (pinned against test/utilities/*.cpp, test/typer/test_path.cpp) ----
  long gto_to_uint64_vec(uint8_t const * codes, long len, long i, uint64_t * out, long cap)
  {
    auto const v = to_uint64_vec(std::vector<uint8_t>(codes, codes + len), static_cast<std::size_t>(i));
    for (long k = 0; k < static_cast<long>(v.size()) && k < cap; ++k)
      out[k] = v[k];
    return static_cast<long>(v.size());
  }

  void gto_hamming1(uint64_t key, uint64_t * out)
  {
    auto const h = hamming1_keys(key);
    std::memcpy(out, h.data(), sizeof(uint64_t) * 96);
  }

  void gto_mismatches_of_last_base(uint64_t key, uint64_t * out)
  {
    auto const h = mismatches_of_last_base(key);
    std::memcpy(out, h.data(), sizeof(uint64_t) * 3);
  }

  void gto_mismatches_of_first_base(uint64_t key, uint64_t * out)
  {
    auto const h = mismatches_of_first_base(key);
    std::memcpy(out, h.data(), sizeof(uint64_t) * 3);
  }

  long gto_get_num_kmers(long len) { return static_cast<long>(get_num_kmers(static_cast<std::size_t>(len))); }
  long gto_ith_kmer_offset(long len, long i) { return static_cast<long>(get_ith_kmer_offset(len, i)); }

  // ---- entry points for the hand-worked vectors of tests/test_oracle_handworked_pairs.py: the functions below have no state
  // but their arguments, so a vector is a row of numbers and the value the reference's text gives for it.
  // A GenotypePaths of the comparison functions as 7 numbers: read_length, longest_path_length, number of paths, mismatches of
  // paths[0], mismatches of the other paths, and per path the number of allele sets without / with the reference allele.
  static GenotypePaths geno_of(uint32_t const * d)
  {
    GenotypePaths g(0, d[0]);
    g.longest_path_length = d[1];
    for (uint32_t i = 0; i < d[2]; ++i)
    {
      Path p;
      p.mismatches = static_cast<uint16_t>(i == 0 ? d[3] : d[4]);
      for (uint32_t k = 0; k < d[5]; ++k)
      {
        p.var_order.push_back(100 + k);
        p.nums.push_back({1});
      }
      for (uint32_t k = 0; k < d[6]; ++k)
      {
        p.var_order.push_back(200 + k);
        p.nums.push_back({0, 1});
      }
      g.paths.push_back(p);
    }
    return g;
  }
  // compare_pair_of_genotype_paths(geno1, geno2) (genotype_paths.cpp:943-974); d = 2 x 7 numbers
  int gto_compare_two(uint32_t const * d)
  {
    GenotypePaths a = geno_of(d), b = geno_of(d + 7);
    return compare_pair_of_genotype_paths(a, b);
  }
  // compare_pair_of_genotype_paths(pair1, pair2) (genotype_paths.cpp:976-1169); d = 4 x 7 numbers: pair1.first, pair1.second,
  // pair2.first, pair2.second
  int gto_compare_pairs(uint32_t const * d)
  {
    GenotypePaths a1 = geno_of(d), a2 = geno_of(d + 7), b1 = geno_of(d + 14), b2 = geno_of(d + 21);
    return compare_pair_of_genotype_paths(std::make_pair(&a1, &a2), std::make_pair(&b1, &b2));
  }
  // the record filter of SV calling (hts_parallel_reader.cpp:528-568)
  int gto_is_good_read(uint32_t flag, int32_t tid, int32_t mtid, int64_t pos, int64_t mpos, uint32_t mapq, uint32_t n_cigar, uint32_t cigar_front,
                       uint32_t cigar_back)
  {
    ReadRecord r;
    r.flag = static_cast<uint16_t>(flag);
    r.tid = tid;
    r.mtid = mtid;
    r.pos = pos;
    r.mpos = mpos;
    r.mapq = static_cast<uint8_t>(mapq);
    r.n_cigar = n_cigar;
    r.cigar_front = cigar_front;
    r.cigar_back = cigar_back;
    return Genotyper::is_good_read(r) ? 1 : 0;
  }
  // are_genotype_paths_good (vcf_writer.cpp:28-60) of a GenotypePaths given as numbers: per path start, end, read_start_index,
  // read_end_index, mismatches; the graph (is_sv_graph) and the options (hq_reads) are the genotyper's
  int gto_paths_good(void * p, long read_length, long n_paths, uint32_t const * d)
  {
    auto & writer = static_cast<GenoHandle *>(p)->g->writer;
    GenotypePaths geno(0, static_cast<std::size_t>(read_length));
    for (long i = 0; i < n_paths; ++i)
    {
      Path path;
      path.start = d[5 * i];
      path.end = d[5 * i + 1];
      path.read_start_index = static_cast<uint16_t>(d[5 * i + 2]);
      path.read_end_index = static_cast<uint16_t>(d[5 * i + 3]);
      path.mismatches = static_cast<uint16_t>(d[5 * i + 4]);
      geno.paths.push_back(path);
    }
    return writer.are_genotype_paths_good(geno) ? 1 : 0;
  }

  // the coverage filter of SV calling (hts_parallel_reader.cpp:594-633) over records given as (position, sample), in order; the
  // first record's position is the bins' origin.  out[i] = 1: the record is let through
  void gto_bin_filter(void * p, long n, int64_t const * pos, int32_t const * sample, uint8_t * out)
  {
    auto & g = *static_cast<GenoHandle *>(p)->g;
    g.bin_counts.clear();
    g.first_pos = n > 0 ? pos[0] : 0;
    for (long i = 0; i < n; ++i)
    {
      ReadRecord r;
      r.pos = pos[i];
      r.sample = sample[i];
      out[i] = g.update_bin_count(r) ? 1 : 0;
    }
  }

  // Haplotype::add_coverage (haplotype.cpp:179-227) over a sequence of alleles, from NO_COVERAGE: the state it ends in
  uint32_t gto_add_coverage(uint16_t const * alleles, long n)
  {
    Haplotype h;
    for (long i = 0; i < n; ++i)
      h.add_coverage(alleles[i]);
    return h.coverage;
  }

  // Graph::get_locations_of_a_position (graph.cpp:1154-1185; how = 2: the position may be a special one and is translated) or
  // get_locations_of_an_actual_position (graph.cpp:931-1029; how = 0 / 1 = is_special) for a path given as its variant orders and
  // the allele sets beside them (bit a of mask k: allele a at var_order[k]), with the path's start and end.  out: type ('R' / 'V'), node, node order, offset per
  // location; returns their number.
  long gto_locations(void * p, uint32_t pos, int how, uint32_t path_start, uint32_t path_end, long n_vars, uint32_t const * var_order, uint32_t const * masks,
                     uint32_t * out, long cap)
  {
    auto const & g = static_cast<Handle *>(p)->graph;
    Path path;
    path.start = path_start; // (a path whose start is its end is "empty", path.cpp:198-201: every allele at its variant orders counts)
    path.end = path_end;
    for (long k = 0; k < n_vars; ++k)
    {
      path.var_order.push_back(var_order[k]);
      std::set<uint16_t> s;
      for (uint16_t a = 0; a < 32; ++a)
        if ((masks[k] >> a) & 1u)
          s.insert(a);
      path.nums.push_back(s);
    }
    std::vector<Location> const locs = how == 2 ? g.get_locations_of_a_position(pos, path) : g.get_locations_of_an_actual_position(pos, path, how == 1);
    long n = 0;
    for (auto const & l : locs)
    {
      if (n < cap)
      {
        out[4 * n] = static_cast<uint32_t>(l.node_type);
        out[4 * n + 1] = l.node_index;
        out[4 * n + 2] = l.node_order;
        out[4 * n + 3] = l.offset;
      }
      ++n;
    }
    return n;
  }

  // Graph::get_labels_forward / get_labels_backward (graph.cpp:1187-1439 / 1441-1701) from one location: the labels (start, end, variant
  // node or 0xFFFFFFFF) of the candidates that tie the fewest mismatches; *max_mismatches goes in as the budget and comes out as their count
  long gto_walk_labels(void * p, int backward, int type, uint32_t node, uint32_t order, uint32_t offset, char const * read, uint32_t * max_mismatches,
                       uint32_t * out, long cap)
  {
    auto const & g = static_cast<Handle *>(p)->graph;
    Location loc;
    loc.node_type = static_cast<char>(type);
    loc.node_index = node;
    loc.node_order = order;
    loc.offset = offset;
    std::string const rd(read);
    std::vector<KmerLabel> const labels = backward ? g.get_labels_backward(loc, rd, *max_mismatches) : g.get_labels_forward(loc, rd, *max_mismatches);
    long n = 0;
    for (auto const & l : labels)
    {
      if (n < cap)
      {
        out[3 * n] = l.start_index;
        out[3 * n + 1] = l.end_index;
        out[3 * n + 2] = l.variant_id;
      }
      ++n;
    }
    return n;
  }

  // the steps of GenotypePaths (genotype_paths.cpp) over paths given as numbers -- per path start, end, read_start_index,
  // read_end_index, mismatches, the number of variant orders, then (order, bit a = allele a) per order -- and back in the same form.
  // op 0: walk_read_ends(seq, arg), 1: walk_read_starts(seq, arg), 2: remove_support_from_read_ends, 3: remove_fully_special_paths,
  // 4: remove_short_paths, 5: remove_paths_with_too_many_mismatches, 6: remove_non_ref_paths_when_read_matches_ref.
  // longest < 0: the longest path's length is taken from the paths.  Returns the number of words written (or needed).
  long gto_paths_op(void * p, int op, char const * seq, int arg, long read_length, long longest, long n_paths, uint32_t const * d, uint32_t * out, long cap)
  {
    auto const & g = static_cast<Handle *>(p)->graph;
    GenotypePaths geno(0, static_cast<std::size_t>(read_length));
    for (long i = 0; i < n_paths; ++i)
    {
      Path path;
      path.start = d[0];
      path.end = d[1];
      path.read_start_index = static_cast<uint16_t>(d[2]);
      path.read_end_index = static_cast<uint16_t>(d[3]);
      path.mismatches = static_cast<uint16_t>(d[4]);
      uint32_t const nv = d[5];
      d += 6;
      for (uint32_t k = 0; k < nv; ++k, d += 2)
      {
        path.var_order.push_back(d[0]);
        std::set<uint16_t> s;
        for (uint16_t a = 0; a < 32; ++a)
          if ((d[1] >> a) & 1u)
            s.insert(a);
        path.nums.push_back(s);
      }
      geno.paths.push_back(path);
    }
    if (longest < 0)
      geno.update_longest_path_size();
    else
      geno.longest_path_length = static_cast<uint32_t>(longest);
    switch (op)
    {
    case 0: geno.walk_read_ends(std::string(seq), arg, g); break;
    case 1: geno.walk_read_starts(std::string(seq), arg, g); break;
    case 2: geno.remove_support_from_read_ends(g); break;
    case 3: geno.remove_fully_special_paths(g); break;
    case 4: geno.remove_short_paths(); break;
    case 5: geno.remove_paths_with_too_many_mismatches(); break;
    case 6: geno.remove_non_ref_paths_when_read_matches_ref(g); break;
    default: return -1;
    }
    long n = 0;
    auto put = [&](uint32_t w)
    {
      if (n < cap)
        out[n] = w;
      ++n;
    };
    put(static_cast<uint32_t>(geno.paths.size()));
    put(geno.longest_path_length);
    for (auto const & path : geno.paths)
    {
      put(path.start);
      put(path.end);
      put(path.read_start_index);
      put(path.read_end_index);
      put(path.mismatches);
      put(static_cast<uint32_t>(path.var_order.size()));
      for (std::size_t k = 0; k < path.var_order.size(); ++k)
      {
        put(path.var_order[k]);
        uint32_t m = 0;
        for (uint16_t a : path.nums[k])
          m |= 1u << a;
        put(m);
      }
    }
    return n;
  }

  // Graph::iterative_dfs (graph.cpp:1703-1754): the walk from every start location (four words each: type, node, order, offset) --
  // or, from one start of type 'U', back from every end location -- keeping the labels of the fewest mismatches
  long gto_walk_between(void * p, long n_starts, uint32_t const * starts, long n_ends, uint32_t const * ends, char const * read, uint32_t * max_mismatches,
                        uint32_t * out, long cap)
  {
    auto const & g = static_cast<Handle *>(p)->graph;
    auto locations = [](long n, uint32_t const * w)
    {
      std::vector<Location> v(static_cast<std::size_t>(n));
      for (long i = 0; i < n; ++i)
      {
        v[i].node_type = static_cast<char>(w[4 * i]);
        v[i].node_index = w[4 * i + 1];
        v[i].node_order = w[4 * i + 2];
        v[i].offset = w[4 * i + 3];
      }
      return v;
    };
    std::vector<KmerLabel> const labels = g.iterative_dfs(locations(n_starts, starts), locations(n_ends, ends), std::string(read), *max_mismatches);
    long n = 0;
    for (auto const & l : labels)
    {
      if (n < cap)
      {
        out[3 * n] = l.start_index;
        out[3 * n + 1] = l.end_index;
        out[3 * n + 2] = l.variant_id;
      }
      ++n;
    }
    return n;
  }

  // make_bi_allelic_call (sample_call.cpp:188-253): d = ambiguous_depth, ref_total_depth, alt_total_depth, alt_proper_pair_depth,
  // then the call's coverage (n_cov values); out = the reduced call's coverage[0], coverage[1], ambiguous_depth, ref_total_depth,
  // alt_total_depth, alt_proper_pair_depth, phred[0..2]
  void gto_make_bi_allelic_call(uint32_t const * d, long n_cov, uint8_t const * phred, long n_phred, long aa, uint32_t * out)
  {
    vcf::SampleCall oc;
    oc.ambiguous_depth = static_cast<uint8_t>(d[0]);
    oc.ref_total_depth = static_cast<uint16_t>(d[1]);
    oc.alt_total_depth = static_cast<uint16_t>(d[2]);
    oc.alt_proper_pair_depth = static_cast<uint8_t>(d[3]);
    for (long i = 0; i < n_cov; ++i)
      oc.coverage.push_back(static_cast<uint16_t>(d[4 + i]));
    oc.phred.assign(phred, phred + n_phred);
    vcf::SampleCall const c = vcf::make_bi_allelic_call(oc, aa);
    out[0] = c.coverage[0];
    out[1] = c.coverage[1];
    out[2] = c.ambiguous_depth;
    out[3] = c.ref_total_depth;
    out[4] = c.alt_total_depth;
    out[5] = c.alt_proper_pair_depth;
    for (int i = 0; i < 3; ++i)
      out[6 + i] = i < static_cast<int>(c.phred.size()) ? c.phred[i] : 0xFFFFu;
  }

  // State of one (haplotype, sample) cell set by hand, for the phase flags (hts_parallel_reader.cpp:782-904 reads nothing else):
  // gt_coverage (n_cov > 0) and the support vector of the connections from allele1 of this haplotype to haplotype hap2
  // (n_support > 0).  Returns the haplotype's variant order (gt.id), -1 when hap / sample are out of range.
  long gto_genotyper_poke(void * p, long hap, long sample, uint16_t const * gt_cov, long n_cov, long allele1, long hap2, uint16_t const * support,
                          long n_support)
  {
    auto & haps = static_cast<GenoHandle *>(p)->g->writer.haplotypes;
    if (hap < 0 || hap >= static_cast<long>(haps.size()) || sample < 0 || sample >= static_cast<long>(haps[hap].hap_samples.size()))
      return -1;
    HapSample & hs = haps[hap].hap_samples[sample];
    if (n_cov > 0)
      hs.gt_coverage.assign(gt_cov, gt_cov + n_cov);
    if (n_support > 0 && allele1 >= 0 && allele1 < static_cast<long>(hs.connections.size()))
      hs.connections[allele1][static_cast<uint16_t>(hap2)] = std::vector<uint16_t>(support, support + n_support);
    return static_cast<long>(haps[hap].id) * 65536 + haps[hap].num;
  }

  // Path(p1,p2) of two id-less labels (test/typer/test_path.cpp:50-65); out: size,start,end,n_var_order,n_nums
  void gto_path_merge_two_ref_labels(uint32_t s1, uint32_t e1, uint32_t rs1, uint32_t re1, uint32_t s2, uint32_t e2, uint32_t rs2,
                                     uint32_t re2, uint32_t * out)
  {
    Graph g;
    Path a(g, KmerLabel(s1, e1), static_cast<uint16_t>(rs1), static_cast<uint16_t>(re1), 0);
    Path b(g, KmerLabel(s2, e2), static_cast<uint16_t>(rs2), static_cast<uint16_t>(re2), 0);
    Path m(a, b);
    out[0] = m.size();
    out[1] = m.start;
    out[2] = m.end;
    out[3] = static_cast<uint32_t>(m.var_order.size());
    out[4] = static_cast<uint32_t>(m.nums.size());
  }

  // bamshrink over one interval (gto_shrink.hpp).  records: the BAM record stream of the whole file (block_size + record, ...);
  // opts: maxFragLen, minNumMatching, is_filtering_mapq0, no_filter_on_coverage, minReadLen, minReadLenMapQ0, minUnpairedReadLen,
  // as_filter_threshold, SUPER_HI_DEPTH, sam_flag_filter, change_read_names; read_num goes in and out (intervals of one call
  // of the reference share it).  Returns the number of bytes of the output stream (-1: malformed input, -2: out too small).
  long gto_bam_shrink(uint8_t const * records, long len, int rid, int interval_begin, int interval_end, long const * opts, double avg_cov,
                      int is_single_contig, long * read_num, uint8_t * out, long cap)
  {
    gto_shrink::Options o;
    o.maxFragLen = static_cast<int>(opts[0]);
    o.minNumMatching = static_cast<int>(opts[1]);
    o.is_filtering_mapq0 = opts[2] != 0;
    o.no_filter_on_coverage = opts[3] != 0;
    o.minReadLen = static_cast<int>(opts[4]);
    o.minReadLenMapQ0 = static_cast<int>(opts[5]);
    o.minUnpairedReadLen = static_cast<int>(opts[6]);
    o.as_filter_threshold = opts[7];
    o.SUPER_HI_DEPTH = opts[8];
    o.sam_flag_filter = static_cast<int>(opts[9]);
    o.change_read_names = opts[10] != 0;
    o.avgCovByReadLen = avg_cov;
    std::vector<gto_shrink::Record> in, res;
    if (!gto_shrink::decode_records(records, static_cast<size_t>(len), in))
      return -1;
    gto_shrink::shrink_interval(o, in, rid, interval_begin, interval_end, *read_num, is_single_contig != 0, res);
    std::vector<uint8_t> bytes;
    for (auto const & r : res)
      gto_shrink::encode_record(r, bytes);
    if (static_cast<long>(bytes.size()) > cap)
      return -2;
    std::memcpy(out, bytes.data(), bytes.size());
    return static_cast<long>(bytes.size());
  }

  long gto_shrink_header(char const * text, char const * chrom, char * out, long cap)
  {
    std::string const h = gto_shrink::single_contig_header(text, chrom);
    if (static_cast<long>(h.size()) + 1 > cap)
      return -2;
    std::memcpy(out, h.c_str(), h.size() + 1);
    return static_cast<long>(h.size());
  }
}
