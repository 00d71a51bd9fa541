// gtx_files.cpp -- graph construction from a FASTA file and a VCF file (host side of the C ABI, include/gtx.h).
//
// Replaces, for graphs without structural-variant alleles, what the reference does in
//   construct_graph            src/graph/constructor.cpp:1597-1777  (region, reference bases, record intake, sort)
//   split_multi_allelic        src/graph/constructor.cpp:1033-1077
//   add_var_record             src/graph/constructor.cpp:1208-1595  (small-variant branch :1493-1588: non-ACGT alts are
//                                                                     dropped, GT_ID / GT_ANTI_HAPLOTYPE become events)
//   GenomicRegion(string)      src/graph/genomic_region.cpp:73-113
//   structural variants        src/graph/constructor.cpp:1079-1207 (transform_sv_records: plain indels of 50 bp and more become
//                              <DEL> / <INS>), :1257-1491 (INFO fields, size defaults, insertions that repeat their
//                              neighbourhood become duplications), add_sv_breakend :312-476, add_sv_deletion :478-514,
//                              add_sv_insertion :515-725, add_sv_duplication :727-871, add_sv_inversion :873-1031,
//                              append_sv_tag_to_node :155-161
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

namespace gtx
{
// gtx_tabix.cpp
bool tabix_start(std::string const & vcf_path, std::string const & chrom, int64_t begin, int64_t end, bool & any, uint64_t & voffset);
gzFile gz_open_at(std::string const & path, uint64_t voffset);
} // namespace gtx

namespace gtx
{
void graph_set_sv_table(gtx_graph * g, std::string table); // gtx_graph.cpp
}

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

struct Rec // one VarRecord in the making (a small variant has one alternative allele, an SV one or two breakpoint alleles)
{
  uint32_t pos = 0;
  bool is_sv = false;
  std::string ref;
  std::vector<std::string> alts;
  std::vector<int64_t> ref_events, alt_events, alt_anti; // (small variants only)
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


// ---- structural variants ---------------------------------------------------------------------------------------
constexpr long EXTRA_SEQUENCE_LENGTH = 152; // constructor.cpp:1437

enum SvType { SV_NONE, SV_DEL, SV_DEL_ALU, SV_DUP, SV_INS, SV_INS_ALU, SV_INV, SV_BND, SV_OTHER };

struct SvFields // gyper::SV (include/graphtyper/graph/sv.hpp:36-58): what the allele synthesis reads and what the calls' post-processing reads later
{
  SvType type = SV_NONE;
  long begin = 0, length = 0, size = 0, end = 0, or_start = -1, or_end = -1;
  long n_clusters = 0, num_merged_svs = -1, related_sv = -1;
  int inv_type = 0; // 1 = INV3, 2 = INV5
  std::string chrom, model = "AGGREGATED", old_variant_id;
  std::string seq, ins_seq, ins_seq_left, ins_seq_right, original_alt;
};

// one line of gtx_graph_sv_table per registered SV
std::string sv_table_line(SvFields const & sv)
{
  static char const * const TYPES[] = {"NOT_SV", "DEL", "DEL_ALU", "DUP", "INS", "INS_ALU", "INV", "BND", "OTHER"};
  auto text = [](std::string const & v) { return v.empty() ? std::string(".") : v; };
  std::string l = TYPES[sv.type];
  l += '\t' + text(sv.chrom);
  for (long v : {sv.begin, sv.length, sv.size, sv.end, sv.n_clusters, sv.num_merged_svs, sv.or_start, sv.or_end, sv.related_sv})
    l += '\t' + std::to_string(v);
  l += '\t' + text(sv.model) + '\t' + text(sv.old_variant_id) + '\t' + (sv.inv_type == 1 ? "INV3" : sv.inv_type == 2 ? "INV5" : ".");
  l += '\t' + text(sv.seq) + "\t." /* HOMSEQ: never read by the constructor */ + '\t' + text(sv.ins_seq) + '\t' + text(sv.ins_seq_left) + '\t' + text(sv.ins_seq_right) +
       '\t' + text(sv.original_alt) + '\n';
  return l;
}

struct SvBuilder
{
  std::string fasta, chr;
  unsigned * n_sv; // Graph::SVs.size(): numbers the SV tags
  std::string err;
  std::string * table = nullptr; // Graph::SVs as text (gtx_graph_sv_table), a line per registered SV

  // read_reference_seq (constructor.cpp:245-257): bases [begin, begin + length) of `contig`, clipped to it
  std::string read(std::string const & contig, long begin, long length)
  {
    std::string out, e;
    if (length <= 0)
      return out;
    if (begin < 0)
    {
      length += begin;
      begin = 0;
    }
    if (length > 0 && !read_fasta_region(fasta, contig, begin, begin + length, out, e) && err.empty())
      err = e;
    return out;
  }
  std::string read(long begin, long length) { return read(chr, begin, length); }

  // read_reference_genome_ends (constructor.cpp:272-293)
  std::string read_ends(long begin, long end, long length)
  {
    if (end - begin > 2 * length)
      return read(begin, length) + read(end - length, length);
    return read(begin, end - begin);
  }

  std::string tag() // append_sv_tag_to_node (constructor.cpp:155-161): numbered by the SVs registered so far
  {
    char t[16];
    std::snprintf(t, sizeof t, "<SV:%07u>", *n_sv);
    return t;
  }
  // graph.SVs.push_back(sv) with the genotyping model set just before; relation +1 / -1: the SV registered next / before is
  // the other breakpoint (sv.related_sv, constructor.cpp:551-560 and the like)
  void register_sv(SvFields sv, char const * model, int relation = 0)
  {
    sv.model = model;
    if (relation != 0)
      sv.related_sv = static_cast<long>(*n_sv) + relation;
    if (table)
      *table += sv_table_line(sv);
    ++*n_sv;
  }

  static char complement(char c) // constructor.cpp:218-243
  {
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return c;
    }
  }
  static std::string revcomp(std::string s)
  {
    std::reverse(s.begin(), s.end());
    for (char & c : s)
      c = complement(c);
    return s;
  }

  // is_similar (constructor.cpp:1353-1394): score of a global alignment (match 1, mismatch -1, gap -1; seqan::globalAlignment
  // with Score<int, Simple>(1, -1, -1)) of at most the first 1000 bases, relative to the longer sequence, at least 0.8
  static bool is_similar(std::string a, std::string b)
  {
    std::size_t constexpr MAX_SIZE = 1000;
    if (a.size() > MAX_SIZE && b.size() > MAX_SIZE)
    {
      a.resize(MAX_SIZE);
      b.resize(MAX_SIZE);
    }
    for (std::string * s : {&a, &b})
      for (char & c : *s)
        c = dna5(c);
    std::size_t const larger = std::max(a.size(), b.size());
    if (larger == 0)
      return false;
    std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
    for (std::size_t j = 0; j <= b.size(); ++j)
      prev[j] = -static_cast<int>(j);
    for (std::size_t i = 1; i <= a.size(); ++i)
    {
      cur[0] = -static_cast<int>(i);
      for (std::size_t j = 1; j <= b.size(); ++j)
        cur[j] = std::max({prev[j - 1] + (a[i - 1] == b[j - 1] ? 1 : -1), prev[j] - 1, cur[j - 1] - 1});
      prev.swap(cur);
    }
    return static_cast<double>(prev[b.size()]) / static_cast<double>(larger) >= 0.8;
  }

  // add_sv_breakend (constructor.cpp:312-476)
  bool breakend(Rec & var, std::string const & alt, SvFields sv)
  {
    sv.original_alt = alt;
    var.ref = read(var.pos, 1);
    auto chrom_of = [&](char c, std::string & name)
    {
      std::size_t const a = alt.find(c), colon = alt.rfind(':');
      if (a == std::string::npos || colon == std::string::npos || colon < a + 1)
        return false;
      name = alt.substr(a + 1, colon - a - 1);
      return true;
    };
    auto position_of = [&](char c, long & pos)
    {
      std::size_t const colon = alt.rfind(':');
      if (colon == std::string::npos)
        return false;
      std::size_t const e = alt.find(c, colon + 1);
      if (e == std::string::npos)
        return false;
      char * endp = nullptr;
      std::string const num = alt.substr(colon + 1, e - colon - 1);
      pos = std::strtol(num.c_str(), &endp, 10);
      return !num.empty() && *endp == '\0';
    };
    std::string bnd, mate;
    long pos = 0;
    std::size_t at = alt.find('[');
    if (at != std::string::npos)
    {
      if (!chrom_of('[', mate) || !position_of('[', pos))
        return false;
      if (at != 0) // case 1: S SNNN[chr:pos[ -- the sequence right of chr:pos follows
      {
        bnd = var.ref + alt.substr(1, at - 1);
        bnd += read(mate, pos, EXTRA_SEQUENCE_LENGTH - static_cast<long>(bnd.size()) + 1);
        bnd += tag();
      }
      else // case 2: S [chr:pos[NNNS -- the reversed sequence left of chr:pos precedes
      {
        std::size_t const second = alt.find('[', at + 1);
        if (second == std::string::npos)
          return false;
        bnd = tag();
        bnd += revcomp(read(mate, pos - 1, EXTRA_SEQUENCE_LENGTH - static_cast<long>(alt.size() - second)));
        bnd += alt.substr(second + 1);
      }
    }
    else
    {
      at = alt.find(']');
      if (at == std::string::npos || !chrom_of(']', mate) || !position_of(']', pos))
        return false;
      if (at == 0) // case 3: S ]chr:pos]NNS -- the sequence up to chr:pos precedes
      {
        std::size_t const second = alt.find(']', at + 1);
        if (second == std::string::npos)
          return false;
        long const len = EXTRA_SEQUENCE_LENGTH - static_cast<long>(alt.size() - second) - 1;
        bnd = tag();
        bnd += read(mate, pos - len, len);
        bnd += alt.substr(second + 1);
      }
      else // case 4: S SNN]chr:pos] -- the reverse complement of the sequence up to chr:pos follows
      {
        bnd = var.ref + alt.substr(1, at - 1);
        long const len = EXTRA_SEQUENCE_LENGTH - static_cast<long>(bnd.size()) + 1;
        bnd += revcomp(read(mate, pos - len, len));
        bnd += tag();
      }
    }
    var.alts.push_back(bnd);
    register_sv(sv, sv.model.c_str());
    return true;
  }

  // add_sv_deletion (constructor.cpp:478-514)
  void deletion(Rec & var, SvFields const & sv)
  {
    var.ref = read(var.pos, 1);
    std::string alt1 = var.ref;
    if (!sv.seq.empty() && sv.seq[0] != '.')
      alt1 += sv.seq;
    else if (!sv.ins_seq.empty() && sv.ins_seq[0] != '.')
      alt1 += sv.ins_seq;
    if (static_cast<long>(alt1.size()) < EXTRA_SEQUENCE_LENGTH + 1)
      alt1 += read(static_cast<long>(var.pos) + static_cast<long>(sv.seq.size()) + sv.size + 1, EXTRA_SEQUENCE_LENGTH + 1 - static_cast<long>(alt1.size()));
    alt1 += tag();
    var.alts.push_back(alt1);
    register_sv(sv, "BREAKPOINT");
  }

  // two breakpoint alleles around an inserted sequence: its first EXTRA_SEQUENCE_LENGTH bases behind the padding base, its
  // last ones in front of the reference that follows (shared by insertions with SEQ and with an origin)
  void two_breakpoints(Rec & var, SvFields const & sv, std::string const & ins, std::string alt1, std::string alt2, long pad_from, bool alt2_has_base)
  {
    if (static_cast<long>(ins.size()) >= EXTRA_SEQUENCE_LENGTH)
    {
      alt1 += ins.substr(0, EXTRA_SEQUENCE_LENGTH);
      alt1 += tag();
      register_sv(sv, "BREAKPOINT1", +1);
      alt2 += tag();
      alt2 += ins.substr(ins.size() - EXTRA_SEQUENCE_LENGTH);
      register_sv(sv, "BREAKPOINT2", -1);
    }
    else
    {
      long padding = EXTRA_SEQUENCE_LENGTH - static_cast<long>(ins.size());
      alt1 += ins;
      alt1 += read(static_cast<long>(var.pos) + 1, padding);
      alt1 += tag();
      register_sv(sv, "BREAKPOINT1", +1);
      alt2 += tag();
      if (alt2_has_base) // (insertion with SEQ: the padding in front of the position and its base, constructor.cpp:577)
        alt2 += read(pad_from - padding, padding + 1);
      else               // (insertion from an origin: not in front of the contig's start, constructor.cpp:636-637)
      {
        padding = std::min<long>(padding, var.pos);
        alt2 += read(static_cast<long>(var.pos) - padding, padding);
      }
      alt2 += ins;
      register_sv(sv, "BREAKPOINT2", -1);
    }
    var.alts.push_back(alt1);
    var.alts.push_back(alt2);
  }

  // add_sv_insertion (constructor.cpp:515-725)
  void insertion(Rec & var, SvFields const & sv, std::string const & vcf_ref)
  {
    var.ref = (!vcf_ref.empty() && vcf_ref[0] != 'N') ? vcf_ref : read(var.pos, 1);
    if (!sv.seq.empty())
    {
      std::string const base = read(var.pos, 1);
      two_breakpoints(var, sv, sv.seq, base, base, var.pos, true);
    }
    else if (sv.or_start != -1 && sv.or_end != -1)
      two_breakpoints(var, sv, read_ends(sv.or_start - 1, sv.or_end, EXTRA_SEQUENCE_LENGTH), read(var.pos, 1), std::string(), var.pos, false);
    else if (!sv.ins_seq_left.empty() || !sv.ins_seq_right.empty())
    {
      std::string const left = sv.ins_seq_left.substr(0, EXTRA_SEQUENCE_LENGTH), right = sv.ins_seq_right.substr(0, EXTRA_SEQUENCE_LENGTH);
      bool const both = left.size() > 1 && !right.empty(); // (only then the two breakpoints name each other)
      if (left.size() > 1)
      {
        var.alts.push_back(var.ref + left + tag());
        register_sv(sv, "BREAKPOINT1", both ? +1 : 0);
      }
      if (!right.empty())
      {
        var.alts.push_back(tag() + right);
        register_sv(sv, "BREAKPOINT2", both ? -1 : 0);
      }
    }
    // (else: the reference does not know how to add the insertion either and the record has no allele)
  }

  // tandem duplication / inversion with both breakpoints (constructor.cpp:743-821, 899-973): `body` is the duplicated or
  // inverted sequence, the first breakpoint allele goes to `first`, the second one to `second`
  void tandem(Rec & first, Rec & second, SvFields const & sv, std::string const & body, std::string head, long first_pos, long second_pos,
              bool tail_gets_ins)
  {
    std::string tail;
    if (static_cast<long>(body.size()) >= EXTRA_SEQUENCE_LENGTH)
    {
      head += body.substr(0, EXTRA_SEQUENCE_LENGTH);
      head += tag();
      register_sv(sv, "BREAKPOINT1", +1);
      tail = tag();
      tail += body.substr(body.size() - EXTRA_SEQUENCE_LENGTH);
      tail += sv.ins_seq;
      register_sv(sv, "BREAKPOINT2", -1);
    }
    else
    {
      long padding = EXTRA_SEQUENCE_LENGTH - static_cast<long>(body.size());
      head += body;
      head += read(first_pos + 1, padding);
      head += tag();
      register_sv(sv, "BREAKPOINT1", +1);
      padding = std::min<long>(padding, second_pos); // (not in front of the contig's start)
      tail = tag();
      tail += read(second_pos - padding + 1, padding);
      tail += body;
      if (tail_gets_ins)
        tail += sv.ins_seq;
      register_sv(sv, "BREAKPOINT2", -1);
    }
    first.alts.push_back(head);
    second.alts.push_back(tail);
  }

  // add_sv_duplication (constructor.cpp:727-871); `extra` receives the record of the other breakpoint
  void duplication(Rec & var, SvFields const & sv, std::vector<Rec> & extra)
  {
    var.ref = read(var.pos, 1);
    if (sv.or_end == -1)
    {
      if (sv.or_start == -1)
      {
        std::string const dup = read_ends(static_cast<long>(var.pos) + 1, static_cast<long>(var.pos) + sv.length + 1, EXTRA_SEQUENCE_LENGTH);
        Rec var2 = var; // (the end of the duplicated sequence stays at the first position)
        var.pos += static_cast<uint32_t>(sv.length);
        var.ref = read(var.pos, 1);
        tandem(var, var2, sv, dup, var.ref + sv.ins_seq, var.pos, var2.pos, false);
        extra.push_back(var2);
      }
      else // only the origin's start is known
      {
        var.alts.push_back(var.ref + sv.ins_seq + read(sv.or_start - 1, EXTRA_SEQUENCE_LENGTH) + tag());
        register_sv(sv, "BREAKPOINT1");
      }
    }
    else // only the origin's end is known
    {
      long const from = std::max<long>(EXTRA_SEQUENCE_LENGTH, sv.or_end);
      var.alts.push_back(tag() + read(from - EXTRA_SEQUENCE_LENGTH, EXTRA_SEQUENCE_LENGTH) + sv.ins_seq);
      register_sv(sv, "BREAKPOINT2");
    }
  }

  // add_sv_inversion (constructor.cpp:873-1031)
  void inversion(Rec & var, SvFields sv, std::vector<Rec> & extra)
  {
    var.ref = read(var.pos, 1);
    if (sv.inv_type == 1)
      sv.or_end = sv.end;
    else if (sv.inv_type == 2)
    {
      sv.or_start = sv.begin;
      sv.begin += sv.size;
      var.pos += static_cast<uint32_t>(sv.size);
      var.ref = read(var.pos, 1);
    }
    if (sv.or_end == -1)
    {
      if (sv.or_start == -1)
      {
        std::string const inv = revcomp(read_ends(static_cast<long>(var.pos) + 1, static_cast<long>(var.pos) + sv.length + 1, EXTRA_SEQUENCE_LENGTH));
        Rec var2 = var;
        var2.pos += static_cast<uint32_t>(sv.length);
        var2.ref = read(var2.pos, 1);
        tandem(var, var2, sv, inv, var.ref + sv.ins_seq, var.pos, var2.pos, true);
        extra.push_back(var2);
      }
      else
      {
        var.alts.push_back(tag() + revcomp(read(sv.or_start - 1, EXTRA_SEQUENCE_LENGTH)) + sv.ins_seq);
        register_sv(sv, "BREAKPOINT2");
      }
    }
    else
    {
      long const from = std::max<long>(EXTRA_SEQUENCE_LENGTH, sv.or_end);
      var.alts.push_back(var.ref + sv.ins_seq + revcomp(read(from - EXTRA_SEQUENCE_LENGTH, EXTRA_SEQUENCE_LENGTH)) + tag());
      register_sv(sv, "BREAKPOINT1");
    }
  }
};

bool parse_int(std::string const & val, long & out) // parse_info_int (constructor.cpp:32-59): the whole value has to be a number
{
  char * endp = nullptr;
  long const v = std::strtol(val.c_str(), &endp, 10);
  if (val.empty() || *endp != '\0')
    return false;
  out = v;
  return true;
}

// transform_sv_records (constructor.cpp:1079-1207): in an SV graph a plain indel whose alleles differ by 50 bases or more
// becomes a symbolic <DEL> / <INS> record.  Returns false when the record is dropped.
bool transform_sv_record(SvBuilder & b, long & pos0, std::string & ref, std::string & alt, std::string & info)
{
  if (alt.empty())
    return false;
  if (pos0 == 0 || alt.find_first_of("<[]") != std::string::npos)
    return true;
  long const size_diff = static_cast<long>(alt.size()) - static_cast<long>(ref.size());
  if (size_diff <= -50)
  {
    std::string seq;
    if (ref[0] != alt[0])
    {
      --pos0;
      ref = b.read(pos0, 1);
      alt = ref + alt;
    }
    if (alt.size() > 1)
      seq = alt.substr(1);
    std::string add = info.empty() ? "" : ";";
    add += "SVTYPE=DEL;SVLEN=" + std::to_string(-size_diff) + ";SVSIZE=" + std::to_string(-size_diff) +
           ";END=" + std::to_string(static_cast<long>(seq.size()) + pos0 + 1 - size_diff);
    if (!seq.empty())
      add += ";SEQ=" + seq;
    ref = ref.substr(0, 1);
    alt = "<DEL>";
    info += add;
  }
  else if (size_diff >= 50)
  {
    std::string seq;
    if (ref[0] != alt[0])
    {
      --pos0;
      ref = b.read(pos0, 1) + ref;
      seq = alt;
    }
    else
      seq = alt.substr(1);
    std::string add = (!info.empty() && info.back() != ';') ? ";" : "";
    add += "SVTYPE=INS;SVLEN=" + std::to_string(size_diff) + ";SVSIZE=" + std::to_string(size_diff) + ";SEQ=" + seq;
    alt = "<INS>";
    info += add;
  }
  return true;
}

// the SV branch of add_var_record (constructor.cpp:1264-1491).  0 = ok (records appended; possibly none), else a status
int add_sv_record(SvBuilder & b, long pos0, std::string const & vcf_ref, std::string const & alt, std::string const & info, std::string const & vcf_id,
                  std::vector<Rec> & recs, std::string & why)
{
  Rec var;
  var.pos = static_cast<uint32_t>(pos0);
  var.is_sv = true;
  SvFields sv;
  sv.begin = pos0 + 1;
  sv.chrom = b.chr;
  sv.old_variant_id = vcf_id; // constructor.cpp:1279-1280
  bool is_a_dup = false;
  for (std::string const & kv : split(info, ';'))
  {
    std::size_t const eq = kv.find('=');
    std::string const key = kv.substr(0, eq), val = eq == std::string::npos ? std::string() : kv.substr(eq + 1);
    bool ok = true;
    if (key == "DUPSVLEN")
      is_a_dup = true;
    if (key == "SVTYPE")
      sv.type = val == "DEL" ? SV_DEL : val == "DEL:ME:ALU" ? SV_DEL_ALU : val == "DUP" ? SV_DUP : val == "INV" ? SV_INV : val == "INS" ? SV_INS
              : val == "INS:ME:ALU" ? SV_INS_ALU : val == "BND" ? SV_BND : SV_OTHER;
    else if (key == "END")
      ok = parse_int(val, sv.end);
    else if (key == "SVSIZE")
      ok = parse_int(val, sv.size);
    else if (key == "SVLEN")
      ok = parse_int(val, sv.length);
    else if (key == "ORSTART")
      ok = parse_int(val, sv.or_start);
    else if (key == "OREND")
      ok = parse_int(val, sv.or_end);
    else if (key == "NCLUSTERS")
      ok = parse_int(val, sv.n_clusters);
    else if (key == "NUM_MERGED_SVS")
      ok = parse_int(val, sv.num_merged_svs);
    else if (key == "SEQ" || key == "SVINSSEQ" || key == "LEFT_SVINSSEQ" || key == "RIGHT_SVINSSEQ" || key == "DUPSVINSSEQ")
    {
      if (!val.empty() && val[0] != '.') // parse_info_str (constructor.cpp:61-77)
        (key == "SEQ" ? sv.seq : key == "LEFT_SVINSSEQ" ? sv.ins_seq_left : key == "RIGHT_SVINSSEQ" ? sv.ins_seq_right : sv.ins_seq) = val;
    }
    else if (key == "INV3")
      sv.inv_type = 1;
    else if (key == "INV5")
      sv.inv_type = 2;
    if (!ok)
    {
      why = "could not parse " + key + "=" + val + " of the INFO field"; // (the reference exits here)
      return GTX_ERR_ARG;
    }
  }
  if (sv.type == SV_NONE)
  {
    why = "structural variant allele '" + alt + "' without SVTYPE";
    return GTX_ERR_ARG;
  }
  if (sv.type == SV_INS && is_a_dup)
    sv.type = SV_DUP; // (Dragen 3.7: small duplications come as SVTYPE=INS)
  if (sv.length < 0)
    sv.length = -sv.length;
  if (sv.type != SV_BND && sv.length == 0)
    sv.length = sv.size ? sv.size : !sv.seq.empty() ? static_cast<long>(sv.seq.size()) : static_cast<long>(sv.ins_seq.size());
  if (sv.size == 0)
    sv.size = sv.length;
  if (sv.end == 0)
    sv.end = sv.begin + sv.size;
  if (sv.type == SV_INS && !sv.seq.empty())
  {
    // an insertion that repeats the sequence next to it is a duplication (constructor.cpp:1352-1434)
    long const n = static_cast<long>(sv.seq.size());
    if (static_cast<long>(var.pos) - 1 - n >= 0)
    {
      std::string const before = b.read(static_cast<long>(var.pos) - 1 - n, n);
      if (SvBuilder::is_similar(before, sv.seq))
      {
        var.pos -= static_cast<uint32_t>(n);
        sv.type = SV_DUP;
      }
    }
    if (sv.type == SV_INS && SvBuilder::is_similar(b.read(static_cast<long>(var.pos) + 1, n), sv.seq))
      sv.type = SV_DUP;
  }
  std::vector<Rec> extra;
  switch (sv.type)
  {
  case SV_BND:
    if (!b.breakend(var, alt, sv))
    {
      why = "invalid breakend allele '" + alt + "'";
      return GTX_ERR_ARG;
    }
    break;
  case SV_DEL:
  case SV_DEL_ALU: b.deletion(var, sv); break;
  case SV_DUP: b.duplication(var, sv, extra); break;
  case SV_INS: b.insertion(var, sv, vcf_ref); break;
  case SV_INV: b.inversion(var, sv, extra); break;
  default: return GTX_OK; // (other types are skipped)
  }
  if (!b.err.empty())
  {
    why = b.err;
    return GTX_ERR_ARG;
  }
  for (Rec & e : extra)
    recs.push_back(std::move(e));
  if (!var.alts.empty())
    recs.push_back(std::move(var));
  return GTX_OK;
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
  std::string sv_table;
  SvBuilder svb{fasta_path, reg.chr, &n_sv, std::string(), &sv_table};
  if (vcf_path && vcf_path[0])
  {
    // With a .tbi / .csi beside a bgzip file the scan starts where the index allows a record of the region and ends behind the
    // region (the reference reads the region's records through tabix: open_tabix / setRegion, constructor.cpp:163-176); without
    // one the whole file is read.
    gzFile z = nullptr;
    bool indexed = false, any = false;
    uint64_t voffset = 0;
    if (gtx::tabix_start(vcf_path, reg.chr, reg.begin, reg.end, any, voffset))
    {
      indexed = true;
      if (any)
      {
        z = gtx::gz_open_at(vcf_path, voffset);
        if (!z) // (the offset is not a member's start: an index that does not belong to this file -- the whole file is read)
          indexed = false;
      }
    }
    if (!indexed)
      z = gzopen(vcf_path, "rb"); // reads plain text as well; bgzip files are concatenated gzip members
    if (!z && !(indexed && !any))
    {
      g_last_error = std::string("gtx_graph_from_files: cannot open VCF ") + vcf_path;
      return GTX_ERR_ARG;
    }
    std::string line;
    std::vector<char> buf(1 << 16);
    bool more = z != nullptr;
    bool seen_contig = false;
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
      {
        if (indexed && seen_contig)
          break; // (an indexed file is sorted: behind the contig)
        continue;
      }
      seen_contig = true;
      long const pos0 = std::atol(col[1].c_str()) - 1;
      if (indexed && pos0 >= reg.end)
        break; // behind the region
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
        long rpos = pos0;
        std::string rref = ref, ralt = alt, rinfo = info;
        if (is_sv_graph && !transform_sv_record(svb, rpos, rref, ralt, rinfo))
          continue; // (constructor.cpp:1666-1676: the record is dropped)
        if (is_sv_alt(ralt))
        {
          if (!is_sv_graph)
          {
            gzclose(z);
            g_last_error = "gtx_graph_from_files: structural variant allele '" + ralt + "' at " + reg.chr + ":" + col[1] + " in a non-SV graph";
            return GTX_ERR_UNSUPPORTED; // (the reference exits: constructor.cpp:1245-1256)
          }
          std::string why;
          int const rc = add_sv_record(svb, rpos, rref, ralt, rinfo, col.size() > 2 ? col[2] : std::string(), recs, why);
          if (rc != GTX_OK)
          {
            gzclose(z);
            g_last_error = "gtx_graph_from_files: " + why + " at " + reg.chr + ":" + col[1];
            return rc;
          }
          continue;
        }
        std::string const & alt_seq = ralt;
        std::string const & ref_seq = rref;
        (void)rinfo;
        if (alt_seq.find_first_not_of("ACGT") != std::string::npos)
          continue; // :1498-1510
        Rec r;
        r.pos = static_cast<uint32_t>(rpos);
        r.ref = ref_seq;
        r.alts.push_back(alt_seq);
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
    if (z)
      gzclose(z);
  }
  // constructor.cpp:1749-1757 (operator< compares positions only; a stable sort keeps file order among equals)
  if (!std::is_sorted(recs.begin(), recs.end(), [](Rec const & a, Rec const & b) { return a.pos < b.pos; }))
    std::stable_sort(recs.begin(), recs.end(), [](Rec const & a, Rec const & b) { return a.pos < b.pos; });
  std::size_t n_alleles = 0;
  for (Rec const & r : recs)
    n_alleles += 1 + r.alts.size();
  std::vector<gtx_allele> alleles(n_alleles);
  std::vector<gtx_record> records(recs.size());
  std::size_t at = 0;
  for (std::size_t i = 0; i < recs.size(); ++i)
  {
    Rec const & r = recs[i];
    records[i] = gtx_record{r.pos, static_cast<uint32_t>(1 + r.alts.size()), &alleles[at], r.is_sv ? 1 : 0};
    alleles[at++] = gtx_allele{r.ref.data(), static_cast<uint32_t>(r.ref.size()), r.ref_events.data(),
                               static_cast<uint32_t>(r.ref_events.size()), nullptr, 0};
    for (std::string const & a : r.alts)
      alleles[at++] = gtx_allele{a.data(), static_cast<uint32_t>(a.size()), r.alt_events.data(), static_cast<uint32_t>(r.alt_events.size()),
                                 r.alt_anti.data(), static_cast<uint32_t>(r.alt_anti.size())};
  }
  long const end = reg.begin + static_cast<long>(refseq.size());
  if (region_begin)
    *region_begin = reg.begin;
  if (region_end)
    *region_end = end;
  // extend_prefix = 1: add_reference_to_record_if_they_have_a_matching_prefix on every record (constructor.cpp:1740-1744)
  // (the region end the reference hands to add_genomic_region is the one of the region string, not clipped to the contig)
  int const rc = gtx_graph_build(refseq.data(), refseq.size(), reg.begin, reg.end, records.data(), static_cast<uint32_t>(records.size()),
                                 add_all_variants, is_sv_graph, 1, out);
  if (rc == GTX_OK)
    gtx::graph_set_sv_table(*out, std::move(sv_table));
  return rc;
}
