// gtx_tabix.cpp -- coordinate index of a BGZF-compressed VCF (host): writing it, and finding where a region starts.
//
// The reference indexes the VCF files it writes with htslib (Vcf::write_tbi_index, /root/reference/src/typer/vcf.cpp:1308-1321:
// tbx_index_build(fn, 0, &tbx_conf_vcf) -> <fn>.tbi, or min_shift 14 -> <fn>.csi with --csi) and reads the variant records
// of a region through such an index when it builds a graph (open_tabix / setRegion, src/graph/constructor.cpp:163-176,
// 1636-1662).  htslib is not in the tree; both formats are public (the tabix and CSI specifications): the index here holds
// what they define -- per contig the bins of the UCSC scheme with the chunks of virtual offsets of their records, the linear
// index of 16 kb windows (.tbi) or a first offset per bin (.csi), the tabix header (VCF preset: contig column 1, position
// column 2, '#' comment lines, end from the REF allele's length or INFO/END) -- so any reader of the formats can use it;
// it does not reproduce htslib's optional merging of sparse bins into their parents (byte equality with htslib's file is
// not claimed and could not be checked here).
#include "gtx_ctx.hpp"
#include "gtx_inflate.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gtx
{
namespace
{
struct Chunk
{
  uint64_t beg, end;
};

struct RefIndex
{
  std::map<uint32_t, std::vector<Chunk>> bins;
  std::vector<uint64_t> linear; // first offset per window of 2^min_shift positions (UINT64_MAX: none yet)
  uint64_t off_beg = UINT64_MAX, off_end = 0, n_records = 0;
  uint32_t last_bin = UINT32_MAX;
};

inline uint32_t bin_first(int level) { return static_cast<uint32_t>(((1ull << (3 * level)) - 1) / 7); }

// hts_reg2bin: the smallest bin that holds [beg, end)
inline uint32_t reg2bin(int64_t beg, int64_t end, int min_shift, int depth)
{
  --end;
  int s = min_shift;
  for (int l = depth; l > 0; --l, s += 3)
    if ((beg >> s) == (end >> s))
      return bin_first(l) + static_cast<uint32_t>(beg >> s);
  return 0;
}

inline int bin_level(uint32_t bin)
{
  int l = 0;
  while (bin >= bin_first(l + 1))
    ++l;
  return l;
}

// One BGZF member after the other, with the virtual offset of every byte handed out
class MemberReader
{
public:
  bool open(std::string const & path)
  {
    fp_ = std::fopen(path.c_str(), "rb");
    return fp_ != nullptr;
  }
  ~MemberReader()
  {
    if (fp_)
      std::fclose(fp_);
  }
  // next line without its '\n' (false at the end); begin / end: virtual offsets of its first byte and of the byte behind its '\n'
  bool line(std::string & out, uint64_t & begin, uint64_t & end)
  {
    out.clear();
    bool any = false;
    for (;;)
    {
      if (at_ == data_.size())
      {
        if (!next_member())
        {
          end = tell();
          return any;
        }
        continue;
      }
      if (!any)
      {
        begin = tell();
        any = true;
      }
      uint8_t const * p = data_.data() + at_;
      uint8_t const * nl = static_cast<uint8_t const *>(std::memchr(p, '\n', data_.size() - at_));
      size_t const take = nl ? static_cast<size_t>(nl - p) : data_.size() - at_;
      out.append(reinterpret_cast<char const *>(p), take);
      at_ += take + (nl ? 1 : 0);
      if (nl)
      {
        end = tell();
        return true;
      }
    }
  }
  bool bad() const { return bad_; }

private:
  // (a position at the end of a member is the same place as offset 0 of the next one; the latter is what indexes hold)
  uint64_t tell() const { return at_ == data_.size() ? static_cast<uint64_t>(next_off_) << 16 : (static_cast<uint64_t>(member_off_) << 16) | at_; }
  bool next_member()
  {
    for (;;)
    {
      long const here = std::ftell(fp_);
      uint8_t h[18];
      size_t const got = std::fread(h, 1, 18, fp_);
      if (got == 0)
        return false;
      if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || h[10] != 6 || h[11] != 0 || h[12] != 'B' || h[13] != 'C')
      {
        bad_ = true; // not BGZF (plain gzip has no member sizes: it cannot be indexed)
        return false;
      }
      size_t const bsize = (h[16] | (h[17] << 8)) + 1u;
      if (bsize < 26)
      {
        bad_ = true;
        return false;
      }
      comp_.resize(bsize - 18 + 8);
      if (std::fread(comp_.data(), 1, bsize - 18, fp_) != bsize - 18)
      {
        bad_ = true;
        return false;
      }
      uint32_t isize;
      std::memcpy(&isize, comp_.data() + bsize - 18 - 4, 4);
      next_off_ = here + static_cast<long>(bsize);
      if (isize == 0)
        continue;
      if (isize > 65536) // (a BGZF member holds at most 64 KB: a damaged ISIZE must not size an allocation)
      {
        bad_ = true;
        return false;
      }
      data_.resize(isize);
      size_t const clen = bsize - 18 - 8;
      bool ok = inflate_raw(comp_.data(), clen, data_.data(), isize);
      if (ok) // (the member's CRC32 holds the library's own decoder to the file: what does not match goes to zlib)
      {
        uint32_t want;
        std::memcpy(&want, comp_.data() + clen, 4);
        ok = crc32_of(data_.data(), isize) == want;
      }
      if (!ok)
      {
        z_stream z{};
        if (inflateInit2(&z, -15) == Z_OK)
        {
          z.next_in = comp_.data();
          z.avail_in = static_cast<uInt>(clen);
          z.next_out = data_.data();
          z.avail_out = isize;
          ok = inflate(&z, Z_FINISH) == Z_STREAM_END && z.avail_out == 0;
          inflateEnd(&z);
          uint32_t want;
          std::memcpy(&want, comp_.data() + clen, 4);
          ok = ok && crc32_of(data_.data(), isize) == want;
        }
      }
      if (!ok)
      {
        bad_ = true;
        return false;
      }
      member_off_ = here;
      at_ = 0;
      return true;
    }
  }
  std::FILE * fp_ = nullptr;
  std::vector<uint8_t> comp_, data_;
  size_t at_ = 0;
  long member_off_ = 0, next_off_ = 0;
  bool bad_ = false;
};

template <class T>
void put(std::string & s, T v)
{
  s.append(reinterpret_cast<char const *>(&v), sizeof(v));
}

// reads a whole (BGZF = multi-member gzip) file
bool read_gz(std::string const & path, std::string & out)
{
  gzFile z = gzopen(path.c_str(), "rb");
  if (!z)
    return false;
  char buf[1 << 16];
  int n;
  while ((n = gzread(z, buf, sizeof buf)) > 0)
    out.append(buf, static_cast<size_t>(n));
  gzclose(z);
  return n == 0;
}

struct Cursor
{
  std::string const & s;
  size_t at = 0;
  bool ok = true;
  template <class T>
  T get()
  {
    T v{};
    if (at + sizeof(T) > s.size())
    {
      ok = false;
      at = s.size();
      return v;
    }
    std::memcpy(&v, s.data() + at, sizeof(T));
    at += sizeof(T);
    return v;
  }
};
} // namespace

// Where to start reading `vcf_path` for records of `chrom` that overlap [begin, end): the smallest chunk start among the
// bins that can hold such a record, not below what the index knows of the region's first window.  false: no usable index
// beside the file (<vcf>.tbi, <vcf>.csi).  any = false: the index knows of no record there.
bool tabix_start(std::string const & vcf_path, std::string const & chrom, int64_t begin, int64_t end, bool & any, uint64_t & voffset)
{
  std::string raw;
  bool csi = false;
  if (!read_gz(vcf_path + ".tbi", raw))
  {
    raw.clear();
    if (!read_gz(vcf_path + ".csi", raw))
      return false;
    csi = true;
  }
  // An index older than its file describes another file: the VCF was written again and the offsets mean nothing (records of the
  // region would silently be left out of the graph).  Not usable -- the caller scans the file.
  {
    struct stat sv{}, si{};
    if (::stat(vcf_path.c_str(), &sv) != 0 || ::stat((vcf_path + (csi ? ".csi" : ".tbi")).c_str(), &si) != 0 || si.st_mtime < sv.st_mtime)
      return false;
  }
  Cursor c{raw};
  char magic[4];
  for (char & m : magic)
    m = c.get<char>();
  int min_shift = 14, depth = 5;
  int32_t n_ref = 0, l_nm = 0;
  if (!csi)
  {
    if (std::memcmp(magic, "TBI\1", 4) != 0)
      return false;
    n_ref = c.get<int32_t>();
    for (int i = 0; i < 6; ++i)
      (void)c.get<int32_t>(); // format, col_seq, col_beg, col_end, meta, skip
    l_nm = c.get<int32_t>();
  }
  else
  {
    if (std::memcmp(magic, "CSI\1", 4) != 0)
      return false;
    min_shift = c.get<int32_t>();
    depth = c.get<int32_t>();
    int32_t const l_aux = c.get<int32_t>();
    if (l_aux < 28 || min_shift < 0 || min_shift > 32 || depth < 0 || depth > 10)
      return false;
    for (int i = 0; i < 6; ++i)
      (void)c.get<int32_t>();
    l_nm = c.get<int32_t>();
    if (l_nm != l_aux - 28)
      return false;
  }
  if (!c.ok || l_nm < 0 || c.at + static_cast<size_t>(l_nm) > raw.size())
    return false;
  int32_t tid = -1, k = 0;
  for (size_t at = c.at, stop = c.at + static_cast<size_t>(l_nm); at < stop; ++k)
  {
    size_t const z = std::min(raw.find('\0', at), stop);
    if (raw.compare(at, z - at, chrom) == 0 && z - at == chrom.size())
      tid = k;
    at = z + 1;
  }
  c.at += static_cast<size_t>(l_nm);
  if (csi)
    n_ref = c.get<int32_t>();
  any = false;
  voffset = UINT64_MAX;
  if (!c.ok || n_ref < 0)
    return false;
  if (tid < 0 || tid >= n_ref)
    return true; // a contig without records
  int64_t const max_pos = 1ll << std::min(62, min_shift + 3 * depth);
  int64_t const last = std::min<int64_t>(std::max<int64_t>(end, begin + 1), max_pos) - 1;
  uint32_t const meta_bin = bin_first(depth + 1) + 1;
  for (int32_t r = 0; c.ok && r <= tid; ++r)
  {
    int32_t const n_bin = c.get<int32_t>();
    uint64_t best = UINT64_MAX, lower = 0;
    int lower_level = -1;
    for (int32_t b = 0; c.ok && b < n_bin; ++b)
    {
      uint32_t const bin = c.get<uint32_t>();
      uint64_t const loffset = csi ? c.get<uint64_t>() : 0;
      int32_t const n_chunk = c.get<int32_t>();
      bool overlaps = false;
      if (r == tid && bin != meta_bin && bin < meta_bin)
      {
        int const l = bin_level(bin);
        int const shift = min_shift + 3 * (depth - l);
        int64_t const kk = static_cast<int64_t>(bin - bin_first(l));
        overlaps = kk >= (begin >> shift) && kk <= (last >> shift);
        if (csi && kk == (begin >> shift) && l > lower_level)
        {
          lower_level = l;
          lower = loffset;
        }
      }
      for (int32_t k2 = 0; c.ok && k2 < n_chunk; ++k2)
      {
        uint64_t const cb = c.get<uint64_t>();
        (void)c.get<uint64_t>();
        if (overlaps && cb < best)
          best = cb;
      }
    }
    if (!csi)
    {
      int32_t const n_intv = c.get<int32_t>();
      for (int32_t i = 0; c.ok && i < n_intv; ++i)
      {
        uint64_t const io = c.get<uint64_t>();
        if (r == tid && i == (begin >> 14))
          lower = io;
      }
    }
    if (c.ok && r == tid && best != UINT64_MAX)
    {
      any = true;
      voffset = std::max(best, lower);
    }
  }
  return c.ok;
}

// A gzFile positioned at a virtual offset of a BGZF file (NULL: could not)
gzFile gz_open_at(std::string const & path, uint64_t voffset)
{
  int const fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0)
    return nullptr;
  // (what the index points at has to be the start of a BGZF member: gzip magic, the extra field with the BC subfield)
  uint8_t h[18];
  if (::lseek(fd, static_cast<off_t>(voffset >> 16), SEEK_SET) < 0 || ::read(fd, h, 18) != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) ||
      h[12] != 'B' || h[13] != 'C' || ::lseek(fd, static_cast<off_t>(voffset >> 16), SEEK_SET) < 0)
  {
    ::close(fd);
    return nullptr;
  }
  gzFile z = gzdopen(fd, "rb");
  if (!z)
  {
    ::close(fd);
    return nullptr;
  }
  char skip[4096];
  for (size_t left = voffset & 0xFFFFu; left;)
  {
    int const n = gzread(z, skip, static_cast<unsigned>(std::min(left, sizeof skip)));
    if (n <= 0)
    {
      gzclose(z);
      return nullptr;
    }
    left -= static_cast<size_t>(n);
  }
  return z;
}
} // namespace gtx

static int tabix_build_body(const char * vcf_gz_path, int min_shift, const char * index_path)
{
  using namespace gtx;
  if (!vcf_gz_path || min_shift < 0 || min_shift > 30)
  {
    g_last_error = "gtx_tabix_build: bad argument";
    return GTX_ERR_ARG;
  }
  bool const csi = min_shift > 0;
  int const shift = csi ? min_shift : 14;
  int const depth = csi ? (31 - min_shift + 2) / 3 : 5; // tbx_index_build: n_lvls = (TBX_MAX_SHIFT - min_shift + 2) / 3
  MemberReader in;
  if (!in.open(vcf_gz_path))
  {
    g_last_error = std::string("gtx_tabix_build: cannot open ") + vcf_gz_path;
    return GTX_ERR_IO;
  }
  std::vector<std::string> names;
  std::vector<RefIndex> refs;
  std::string line;
  uint64_t off_beg = 0, off_end = 0;
  int32_t tid = -1;
  int64_t last_beg = -1;
  while (in.line(line, off_beg, off_end))
  {
    if (line.empty() || line[0] == '#')
      continue;
    // columns 1 (contig), 2 (position), 4 (REF), 8 (INFO)
    size_t c0 = line.find('\t');
    if (c0 == std::string::npos)
    {
      g_last_error = std::string("gtx_tabix_build: a line without columns in ") + vcf_gz_path;
      return GTX_ERR_ARG;
    }
    if (tid < 0 || line.compare(0, c0, names[static_cast<size_t>(tid)]) != 0 || names[static_cast<size_t>(tid)].size() != c0)
    {
      std::string const name = line.substr(0, c0);
      if (std::find(names.begin(), names.end(), name) != names.end())
      {
        g_last_error = "gtx_tabix_build: the records of contig " + name + " are not in one block (the file is not sorted)";
        return GTX_ERR_ARG;
      }
      names.push_back(name);
      refs.emplace_back();
      tid = static_cast<int32_t>(names.size()) - 1;
      last_beg = -1;
    }
    size_t col = 1, at = c0 + 1;
    int64_t beg = std::atoll(line.c_str() + at) - 1, end = -1;
    while (col < 8 && at != std::string::npos)
    {
      size_t const tab = line.find('\t', at);
      size_t const stop = tab == std::string::npos ? line.size() : tab;
      if (col == 3) // REF
        end = beg + static_cast<int64_t>(stop - at);
      else if (col == 7) // INFO: END= at its start or behind a ';'
      {
        std::string const info = line.substr(at, stop - at);
        bool const in_front = info.compare(0, 4, "END=") == 0;
        size_t e = in_front ? 4 : info.find(";END=");
        if (e != std::string::npos)
        {
          if (!in_front)
            e += 5;
          if (e < info.size() && info[e] != '.')
          {
            long long const v = std::atoll(info.c_str() + e);
            if (v > beg)
              end = v;
          }
        }
      }
      at = tab == std::string::npos ? tab : tab + 1;
      ++col;
    }
    if (beg < 0 || end <= beg)
      end = beg + 1;
    if (beg < 0 || beg < last_beg)
    {
      g_last_error = std::string("gtx_tabix_build: positions are not sorted in ") + vcf_gz_path;
      return GTX_ERR_ARG;
    }
    if (end > (1ll << (shift + 3 * depth)))
    {
      g_last_error = "gtx_tabix_build: a position beyond what the index geometry holds (a .csi with a larger min_shift does)";
      return GTX_ERR_ARG;
    }
    last_beg = beg;
    RefIndex & r = refs[static_cast<size_t>(tid)];
    uint32_t const bin = reg2bin(beg, end, shift, depth);
    auto & chunks = r.bins[bin];
    if (bin == r.last_bin && !chunks.empty() && chunks.back().end == off_beg)
      chunks.back().end = off_end; // the run of records of one bin goes on
    else
      chunks.push_back(Chunk{off_beg, off_end});
    r.last_bin = bin;
    size_t const w1 = static_cast<size_t>((end - 1) >> shift);
    if (r.linear.size() <= w1)
      r.linear.resize(w1 + 1, UINT64_MAX);
    for (size_t w = static_cast<size_t>(beg >> shift); w <= w1; ++w)
      if (r.linear[w] == UINT64_MAX)
        r.linear[w] = off_beg;
    r.off_beg = std::min(r.off_beg, off_beg);
    r.off_end = std::max(r.off_end, off_end);
    ++r.n_records;
  }
  if (in.bad())
  {
    g_last_error = std::string("gtx_tabix_build: ") + vcf_gz_path + " is not a readable BGZF file (bgzip, not gzip)";
    return GTX_ERR_IO;
  }
  std::string nm;
  for (auto const & n : names)
    nm.append(n.c_str(), n.size() + 1);
  std::string out;
  auto header = [&](std::string & s)
  {
    put<int32_t>(s, 2); // format: VCF
    put<int32_t>(s, 1); // col_seq
    put<int32_t>(s, 2); // col_beg
    put<int32_t>(s, 0); // col_end
    put<int32_t>(s, '#');
    put<int32_t>(s, 0); // skip
    put<int32_t>(s, static_cast<int32_t>(nm.size()));
    s += nm;
  };
  if (!csi)
  {
    out.append("TBI\1", 4);
    put<int32_t>(out, static_cast<int32_t>(names.size()));
    header(out);
  }
  else
  {
    out.append("CSI\1", 4);
    put<int32_t>(out, shift);
    put<int32_t>(out, depth);
    put<int32_t>(out, static_cast<int32_t>(28 + nm.size()));
    header(out);
    put<int32_t>(out, static_cast<int32_t>(names.size()));
  }
  uint32_t const meta_bin = bin_first(depth + 1) + 1;
  for (RefIndex & r : refs)
  {
    // windows without a record of their own take the next one's offset (what a reader may skip to)
    for (size_t w = r.linear.size(); w-- > 1;)
      if (r.linear[w - 1] == UINT64_MAX)
        r.linear[w - 1] = r.linear[w];
    put<int32_t>(out, static_cast<int32_t>(r.bins.size() + 1));
    for (auto const & b : r.bins)
    {
      put<uint32_t>(out, b.first);
      if (csi)
      {
        // the offset of the first record that overlaps the bin's first window
        int const l = bin_level(b.first);
        size_t const w = static_cast<size_t>(b.first - bin_first(l)) << (3 * (depth - l));
        put<uint64_t>(out, w < r.linear.size() ? r.linear[w] : 0);
      }
      put<int32_t>(out, static_cast<int32_t>(b.second.size()));
      for (Chunk const & c : b.second)
      {
        put<uint64_t>(out, c.beg);
        put<uint64_t>(out, c.end);
      }
    }
    put<uint32_t>(out, meta_bin); // the pseudo-bin: where the contig's records lie, how many there are
    if (csi)
      put<uint64_t>(out, 0);
    put<int32_t>(out, 2);
    put<uint64_t>(out, r.off_beg);
    put<uint64_t>(out, r.off_end);
    put<uint64_t>(out, r.n_records);
    put<uint64_t>(out, 0);
    if (!csi)
    {
      put<int32_t>(out, static_cast<int32_t>(r.linear.size()));
      for (uint64_t v : r.linear)
        put<uint64_t>(out, v);
    }
  }
  put<uint64_t>(out, 0); // n_no_coor
  uint64_t n = 0;
  std::vector<uint8_t> packed(out.size() + out.size() / 8 + (out.size() / 0xff00u + 2) * 64);
  if (gtx_bgzf_compress(out.data(), out.size(), -1, 1, packed.data(), packed.size(), &n) != GTX_OK)
    return GTX_ERR_IO;
  std::string const path = index_path && index_path[0] ? std::string(index_path) : std::string(vcf_gz_path) + (csi ? ".csi" : ".tbi");
  std::FILE * fp = std::fopen(path.c_str(), "wb");
  if (!fp || std::fwrite(packed.data(), 1, n, fp) != n || std::fclose(fp) != 0)
  {
    g_last_error = "gtx_tabix_build: cannot write " + path;
    return GTX_ERR_IO;
  }
  return GTX_OK;
}

static int tabix_start_body(const char * vcf_gz_path, const char * chrom, int64_t begin, int64_t end, uint64_t * voffset, int * any)
{
  if (!vcf_gz_path || !chrom || !voffset || !any || begin < 0)
    return GTX_ERR_ARG;
  bool a = false;
  uint64_t v = 0;
  if (!gtx::tabix_start(vcf_gz_path, chrom, begin, end, a, v))
  {
    gtx::g_last_error = std::string("gtx_tabix_start: no usable .tbi / .csi beside ") + vcf_gz_path;
    return GTX_ERR_IO;
  }
  *any = a ? 1 : 0;
  *voffset = a ? v : 0;
  return GTX_OK;
}

// (nothing a damaged file can provoke -- an allocation that fails, a container's range check -- may cross the C boundary)
extern "C" int gtx_tabix_build(const char * vcf_gz_path, int min_shift, const char * index_path)
{
  try
  {
    return tabix_build_body(vcf_gz_path, min_shift, index_path);
  }
  catch (std::exception const & e)
  {
    gtx::g_last_error = std::string("gtx_tabix_build: ") + e.what();
    return GTX_ERR_IO;
  }
}

extern "C" int gtx_tabix_start(const char * vcf_gz_path, const char * chrom, int64_t begin, int64_t end, uint64_t * voffset, int * any)
{
  try
  {
    return tabix_start_body(vcf_gz_path, chrom, begin, end, voffset, any);
  }
  catch (std::exception const & e)
  {
    gtx::g_last_error = std::string("gtx_tabix_start: ") + e.what();
    return GTX_ERR_IO;
  }
}
