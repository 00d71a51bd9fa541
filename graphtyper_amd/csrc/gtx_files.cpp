// gtx_files.cpp -- graph construction from a FASTA file and a VCF file (host side of the C ABI, include/gtx.h).
//
// Replaces, for graphs without structural-variant alleles, what the reference does in
//   construct_graph            src/graph/constructor.cpp:1597-1777  (region, reference bases, record intake, sort)
//   split_multi_allelic        src/graph/constructor.cpp:1033-1077
//   add_var_record             src/graph/constructor.cpp:1208-1595  (small-variant branch :1493-1588: non-ACGT alts are
//                                                                     dropped, GT_ID / GT_ANTI_HAPLOTYPE become events)
//   GenomicRegion(string)      src/graph/genomic_region.cpp:73-113
//   SV deletions               src/graph/constructor.cpp:1257-1349 (INFO fields, size defaults), add_sv_deletion :478-514,
//                              append_sv_tag_to_node :155-161.  The other SV types (DUP, INV, INS, BND) are refused.
// and then hands the records to the builder behind gtx_graph_build (record merging, node emission).
// Own parsers: the reference reads FASTA through seqan's FaiIndex (bases arrive as Dna5: anything but ACGT is N) and VCF
// lines through seqan / tabix; here the FASTA is read through its .fai when present (else scanned) and the VCF -- plain
// or gzip/bgzip -- is scanned line by line with zlib.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gtx.h"
#include "gtx_ctx.hpp"

namespace
{
struct Region
{
  std::string chr = "N/A";
  long begin = 0;              // 0-based
  long end = 0xFFFFFFFFl;      // AS_LONG_AS_POSSIBLE (constants.hpp)
};

// genomic_region.cpp:73-113
bool parse_region(std::string const & s, Region & r, std::string & err)
{
  if (s.empty() || s == ".")
    return true;
  std::size_t const colon = s.find(':');
  try
  {
    if (colon == std::string::npos)
      r.chr = s;
    else
    {
      r.chr = s.substr(0, colon);
      std::size_t const dash = s.find('-', colon + 1);
      if (dash == std::string::npos)
        r.begin = std::stol(s.substr(colon + 1));
      else
      {
        r.begin = std::stol(s.substr(colon + 1, dash - colon - 1));
        r.end = std::stol(s.substr(dash + 1));
      }
    }
  }
  catch (std::exception const &)
  {
    err = "cannot parse region '" + s + "'";
    return false;
  }
  if (r.begin != 0)
    --r.begin; // to 0-based
  return true;
}

char dna5(char c) // seqan::Dna5 conversion
{
  switch (c)
  {
  case 'A': case 'a': return 'A';
  case 'C': case 'c': return 'C';
  case 'G': case 'g': return 'G';
  case 'T': case 't': return 'T';
  default: return 'N';
  }
}

// bases [begin, end) of contig `chr` (clipped to the contig); uses <fasta>.fai when it exists
bool read_fasta_region(std::string const & path, std::string const & chr, long begin, long end, std::string & out, std::string & err)
{
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open())
  {
    err = "cannot open FASTA " + path;
    return false;
  }
  out.clear();
  std::ifstream fai(path + ".fai");
  if (fai.is_open())
  {
    std::string name;
    long length, offset, line_bases, line_width;
    for (std::string line; std::getline(fai, line);)
    {
      std::istringstream ss(line);
      if (!(ss >> name >> length >> offset >> line_bases >> line_width) || name != chr)
        continue;
      if (line_bases <= 0 || line_width < line_bases)
      {
        err = "malformed FASTA index entry for " + chr;
        return false;
      }
      end = std::min(end, length);
      if (begin >= end)
        return true;
      long const first = offset + (begin / line_bases) * line_width + begin % line_bases;
      long const last = offset + ((end - 1) / line_bases) * line_width + (end - 1) % line_bases;
      std::string raw(static_cast<std::size_t>(last - first + 1), '\0');
      f.seekg(first);
      f.read(&raw[0], static_cast<std::streamsize>(raw.size()));
      if (f.gcount() != static_cast<std::streamsize>(raw.size()))
      {
        err = "FASTA shorter than its index says";
        return false;
      }
      out.reserve(static_cast<std::size_t>(end - begin));
      for (char c : raw)
        if (c != '\n' && c != '\r')
          out.push_back(dna5(c));
      return true;
    }
    err = "contig " + chr + " not found in " + path + ".fai";
    return false;
  }
  // no index: scan
  bool in_contig = false, found = false;
  long at = 0;
  for (std::string line; std::getline(f, line);)
  {
    if (!line.empty() && line.back() == '\r')
      line.pop_back();
    if (!line.empty() && line[0] == '>')
    {
      if (in_contig)
        break;
      std::string name = line.substr(1, line.find_first_of(" \t") == std::string::npos ? std::string::npos : line.find_first_of(" \t") - 1);
      in_contig = name == chr;
      found = found || in_contig;
      continue;
    }
    if (!in_contig)
      continue;
    for (char c : line)
    {
      if (at >= begin && at < end)
        out.push_back(dna5(c));
      ++at;
    }
    if (at >= end)
      break;
  }
  if (!found)
  {
    err = "contig " + chr + " not found in " + path;
    return false;
  }
  return true;
}

struct Rec // one biallelic VarRecord in the making
{
  uint32_t pos;
  bool is_sv = false;
  std::string ref, alt;
  std::vector<int64_t> ref_events, alt_events, alt_anti;
};

std::vector<std::string> split(std::string const & s, char sep)
{
  std::vector<std::string> out;
  std::size_t a = 0;
  for (;;)
  {
    std::size_t const b = s.find(sep, a);
    out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
    if (b == std::string::npos)
      break;
    a = b + 1;
  }
  return out;
}

bool is_sv_alt(std::string const & alt) // constructor.cpp:1236-1262
{
  if (alt.size() < 5)
    return false;
  return alt.find_first_of("<[]") != std::string::npos;
}
} // namespace

extern "C" int gtx_graph_from_files(const char * fasta_path, const char * vcf_path, const char * region, int add_all_variants,
                                    int is_sv_graph, gtx_graph ** out, int64_t * region_begin, int64_t * region_end)
{
  using gtx::g_last_error;
  if (!fasta_path || !region || !out)
  {
    g_last_error = "gtx_graph_from_files: NULL argument";
    return GTX_ERR_ARG;
  }
  Region reg;
  std::string err, refseq;
  if (!parse_region(region, reg, err) || !read_fasta_region(fasta_path, reg.chr, reg.begin, reg.end, refseq, err))
  {
    g_last_error = "gtx_graph_from_files: " + err;
    return GTX_ERR_ARG;
  }
  if (refseq.empty())
  {
    g_last_error = std::string("gtx_graph_from_files: no reference bases for region ") + region; // constructor.cpp:1621-1625
    return GTX_ERR_ARG;
  }
  std::vector<Rec> recs;
  unsigned n_sv = 0; // Graph::SVs.size(): numbers the SV tags
  if (vcf_path && vcf_path[0])
  {
    gzFile z = gzopen(vcf_path, "rb"); // reads plain text as well; bgzip files are concatenated gzip members
    if (!z)
    {
      g_last_error = std::string("gtx_graph_from_files: cannot open VCF ") + vcf_path;
      return GTX_ERR_ARG;
    }
    std::string line;
    std::vector<char> buf(1 << 16);
    bool more = true;
    while (more)
    {
      line.clear();
      for (;;) // one line of any length
      {
        if (!gzgets(z, buf.data(), static_cast<int>(buf.size())))
        {
          more = false;
          break;
        }
        line += buf.data();
        if (!line.empty() && line.back() == '\n')
          break;
      }
      while (!line.empty() && (line.back() == '\n' || line.back() == '\r'))
        line.pop_back();
      if (line.empty() || line[0] == '#')
        continue;
      std::vector<std::string> const col = split(line, '\t');
      if (col.size() < 5 || col[0] != reg.chr) // (the reference reads the contig's records through tabix)
        continue;
      long const pos0 = std::atol(col[1].c_str()) - 1;
      std::string const & ref = col[3];
      // constructor.cpp:1660-1662: the record has to lie inside the region
      if (pos0 < reg.begin || pos0 + static_cast<long>(ref.size()) > reg.end)
        continue;
      if (ref.empty() || col[4].empty())
        continue; // split_multi_allelic :1037-1047
      std::string const info = col.size() > 7 ? col[7] : std::string();
      for (std::string const & alt : split(col[4], ','))
      {
        if (alt.empty() || alt[0] == '.')
          continue; // :1064-1068
        if (is_sv_alt(alt) && is_sv_graph && alt.compare(0, 4, "<DEL") == 0)
        {
          // a deletion: the reference allele is the base at the position, the alternative allele that base, any inserted
          // sequence, and the reference behind the deleted stretch up to EXTRA_SEQUENCE_LENGTH + 1 characters, closed by
          // the SV tag that stops walks and k-mers at the allele's end
          constexpr std::size_t EXTRA_SEQUENCE_LENGTH = 152; // constructor.cpp:1437
          std::string sv_type, seq, ins_seq;
          long sv_size = 0, sv_len = 0;
          for (std::string const & kv : split(info, ';'))
          {
            std::size_t const eq = kv.find('=');
            if (eq == std::string::npos)
              continue;
            std::string const key = kv.substr(0, eq), val = kv.substr(eq + 1);
            if (key == "SVTYPE")
              sv_type = val;
            else if (key == "SVSIZE")
              sv_size = std::atol(val.c_str());
            else if (key == "SVLEN")
              sv_len = std::atol(val.c_str());
            else if (key == "SEQ")
              seq = val;
            else if (key == "SVINSSEQ")
              ins_seq = val;
          }
          if (sv_type != "DEL" && sv_type != "DEL:ME:ALU")
          {
            gzclose(z);
            g_last_error = "gtx_graph_from_files: allele '" + alt + "' at " + reg.chr + ":" + col[1] + " without SVTYPE=DEL";
            return GTX_ERR_ARG;
          }
          if (sv_len < 0)
            sv_len = -sv_len; // :1331-1332
          if (sv_len == 0)    // :1335-1346
            sv_len = sv_size ? sv_size : seq.size() ? static_cast<long>(seq.size()) : static_cast<long>(ins_seq.size());
          if (sv_size == 0)
            sv_size = sv_len; // :1349-1350
          Rec r;
          r.pos = static_cast<uint32_t>(pos0);
          r.is_sv = true;
          std::string piece;
          if (!read_fasta_region(fasta_path, reg.chr, pos0, pos0 + 1, r.ref, err) || r.ref.size() != 1)
          {
            gzclose(z);
            g_last_error = "gtx_graph_from_files: no reference base at " + reg.chr + ":" + col[1];
            return GTX_ERR_ARG;
          }
          r.alt = r.ref;
          if (!seq.empty() && seq[0] != '.')
            r.alt += seq;
          else if (!ins_seq.empty() && ins_seq[0] != '.')
            r.alt += ins_seq;
          if (r.alt.size() < EXTRA_SEQUENCE_LENGTH + 1)
          {
            long const from = pos0 + static_cast<long>(seq.size()) + sv_size + 1;
            if (!read_fasta_region(fasta_path, reg.chr, from, from + static_cast<long>(EXTRA_SEQUENCE_LENGTH + 1 - r.alt.size()), piece, err))
            {
              gzclose(z);
              g_last_error = "gtx_graph_from_files: " + err;
              return GTX_ERR_ARG;
            }
            r.alt += piece; // (clipped at the contig's end, like seqan's readRegion)
          }
          char tag[16];
          std::snprintf(tag, sizeof tag, "<SV:%07u>", n_sv++);
          r.alt += tag;
          recs.push_back(std::move(r));
          continue;
        }
        if (is_sv_alt(alt))
        {
          gzclose(z);
          g_last_error = "gtx_graph_from_files: structural variant allele '" + alt + "' at " + reg.chr + ":" + col[1] +
                         (is_sv_graph ? " (of the SV types only deletions are built by this library yet)" : " in a non-SV graph");
          return GTX_ERR_UNSUPPORTED;
        }
        if (alt.find_first_not_of("ACGT") != std::string::npos)
          continue; // :1498-1510
        Rec r;
        r.pos = static_cast<uint32_t>(pos0);
        r.ref = ref;
        r.alt = alt;
        for (std::string const & kv : split(info, ';')) // :1540-1585
        {
          std::size_t const eq = kv.find('=');
          if (eq == std::string::npos)
            continue;
          std::string const key = kv.substr(0, eq), val = kv.substr(eq + 1);
          if (key == "GT_ID")
          {
            long const id = std::atol(val.c_str());
            r.ref_events.push_back(-id);
            r.alt_events.push_back(id);
          }
          else if (key == "GT_ANTI_HAPLOTYPE")
            for (std::string const & v : split(val, ','))
              r.alt_anti.push_back(std::atol(v.c_str()));
        }
        recs.push_back(std::move(r));
      }
    }
    gzclose(z);
  }
  // constructor.cpp:1749-1757 (operator< compares positions only; a stable sort keeps file order among equals)
  if (!std::is_sorted(recs.begin(), recs.end(), [](Rec const & a, Rec const & b) { return a.pos < b.pos; }))
    std::stable_sort(recs.begin(), recs.end(), [](Rec const & a, Rec const & b) { return a.pos < b.pos; });
  std::vector<gtx_allele> alleles(2 * recs.size());
  std::vector<gtx_record> records(recs.size());
  for (std::size_t i = 0; i < recs.size(); ++i)
  {
    Rec const & r = recs[i];
    alleles[2 * i] = gtx_allele{r.ref.data(), static_cast<uint32_t>(r.ref.size()), r.ref_events.data(),
                                static_cast<uint32_t>(r.ref_events.size()), nullptr, 0};
    alleles[2 * i + 1] = gtx_allele{r.alt.data(), static_cast<uint32_t>(r.alt.size()), r.alt_events.data(),
                                    static_cast<uint32_t>(r.alt_events.size()), r.alt_anti.data(), static_cast<uint32_t>(r.alt_anti.size())};
    records[i] = gtx_record{r.pos, 2, &alleles[2 * i], r.is_sv ? 1 : 0};
  }
  long const end = reg.begin + static_cast<long>(refseq.size());
  if (region_begin)
    *region_begin = reg.begin;
  if (region_end)
    *region_end = end;
  // extend_prefix = 1: add_reference_to_record_if_they_have_a_matching_prefix on every record (constructor.cpp:1740-1744)
  // (the region end the reference hands to add_genomic_region is the one of the region string, not clipped to the contig)
  return gtx_graph_build(refseq.data(), refseq.size(), reg.begin, reg.end, records.data(), static_cast<uint32_t>(records.size()),
                         add_all_variants, is_sv_graph, 1, out);
}
