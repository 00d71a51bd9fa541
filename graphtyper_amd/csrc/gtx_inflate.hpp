// gtx_inflate.hpp -- raw DEFLATE (RFC 1951) decoder for BGZF members (host).
//
// Reading a BAM file is nine tenths inflate (gtx_bam.cpp), and with 16+ reader threads the pipeline's throughput IS the
// host's inflate rate.  zlib's inflate is a resumable state machine that decodes one symbol per table walk with a 32-bit
// bit buffer; a BGZF member is a whole, small (<= 64 KB in, <= 64 KB out) stream whose output size is known before the first
// bit is read, so this decoder is built for exactly that case:
//   * 64-bit bit buffer refilled with one unaligned 8-byte load (the caller guarantees 8 readable bytes behind the stream:
//     a BGZF member ends with CRC32 + ISIZE);
//   * one table per alphabet with an 11-bit (literal/length) / 8-bit (distance) first level holding everything decoding
//     needs in one 32-bit entry -- literal value, or length/distance base with its number of extra bits -- and second-level
//     tables only for codes longer than that;
//   * a fast loop that runs while 8 input bytes and 258 + 8 output bytes of room remain (up to three literals per refill,
//     matches copied 8 bytes at a time when the distance allows), and a careful loop for the rest;
//   * the whole output lies in one buffer: matches reach back into it directly, no window.
// Any violation of the format (bad block type, over-subscribed or incomplete code, distance in front of the output,
// output longer or shorter than expected, input exhausted) returns false and leaves the output undefined; nothing is
// read beyond in + in_len + 8 or written beyond out + out_len.
#pragma once
#include <cstdint>
#include <cstring>

namespace gtx
{
namespace inflate_detail
{
constexpr unsigned LITLEN_BITS = 11, DIST_BITS = 8;
constexpr unsigned LITLEN_SIZE = (1u << LITLEN_BITS) + 1024, DIST_SIZE = (1u << DIST_BITS) + 512; // first level + room for the second

// entry: bits 0..4 code length to consume (second-level entries: the part behind the first level)
//        bits 5..7 kind: 0 literal, 1 length, 2 end of block, 3 second-level pointer, 4 invalid
//        bits 8..12 extra bits (length / distance) or second-level table bits (pointer)
//        bits 16..31 literal / base / second-level offset
enum : uint32_t { K_LIT = 0u << 5, K_LEN = 1u << 5, K_EOB = 2u << 5, K_SUB = 3u << 5, K_BAD = 4u << 5, K_MASK = 7u << 5 };

inline uint32_t entry(uint32_t kind, uint32_t value, uint32_t extra, uint32_t len) { return (value << 16) | (extra << 8) | kind | len; }

static uint16_t const LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static uint8_t const LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static uint16_t const DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static uint8_t const DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t symbol_entry(bool litlen, unsigned sym, unsigned len)
{
  if (!litlen)
    return sym < 30 ? entry(K_LEN, DIST_BASE[sym], DIST_EXTRA[sym], len) : entry(K_BAD, 0, 0, len);
  if (sym < 256)
    return entry(K_LIT, sym, 0, len);
  if (sym == 256)
    return entry(K_EOB, 0, 0, len);
  return sym < 286 ? entry(K_LEN, LEN_BASE[sym - 257], LEN_EXTRA[sym - 257], len) : entry(K_BAD, 0, 0, len);
}

// Canonical Huffman code of `n` symbols with the given lengths (0 = unused, <= 15) into `table` (first level of `bits` bits,
// second levels behind it, `cap` entries in all).  false: over-subscribed, incomplete (but for a code of one symbol of length 1,
// which zlib takes as well), or more second-level entries than `cap` holds (the caller then leaves the member to zlib).
inline bool build_table(uint8_t const * lens, unsigned n, bool litlen, unsigned bits, uint32_t * table, unsigned cap)
{
  unsigned count[16] = {0};
  for (unsigned i = 0; i < n; ++i)
    ++count[lens[i]];
  for (unsigned i = 0; i < (1u << bits); ++i)
    table[i] = entry(K_BAD, 0, 0, 1);
  if (count[0] == n) // no symbol at all (a block without matches may say so of its distances)
    return !litlen;
  long left = 1;
  unsigned max_len = 0;
  for (unsigned l = 1; l <= 15; ++l)
  {
    left = (left << 1) - static_cast<long>(count[l]);
    if (left < 0)
      return false;
    if (count[l])
      max_len = l;
  }
  if (left > 0 && max_len != 1)
    return false;
  // symbols sorted by (length, value)
  uint16_t sorted[288];
  unsigned offs[16];
  offs[1] = 0;
  for (unsigned l = 1; l < 15; ++l)
    offs[l + 1] = offs[l] + count[l];
  for (unsigned i = 0; i < n; ++i)
    if (lens[i])
      sorted[offs[lens[i]]++] = static_cast<uint16_t>(i);
  // codes in increasing order; the tables are indexed by the code's bits in the order they arrive: reversed
  unsigned rem[16];
  for (unsigned l = 0; l < 16; ++l)
    rem[l] = count[l];
  unsigned code = 0, k = 0, next_sub = 1u << bits;
  unsigned sub_prefix = ~0u, sub_bits = 0, sub_at = 0;
  for (unsigned l = 1; l <= max_len; ++l, code <<= 1)
    for (unsigned c = 0; c < count[l]; ++c, ++code, ++k)
    {
      unsigned rev = 0;
      for (unsigned b = 0; b < l; ++b)
        rev |= ((code >> b) & 1u) << (l - 1 - b);
      if (l <= bits)
      {
        uint32_t const e = symbol_entry(litlen, sorted[k], l);
        for (unsigned i = rev; i < (1u << bits); i += 1u << l)
          table[i] = e;
        --rem[l];
        continue;
      }
      unsigned const prefix = rev & ((1u << bits) - 1);
      if (prefix != sub_prefix)
      {
        // a new second level: the codes still to come fill the space below this prefix in order -- it is as wide as the
        // longest of them that fits
        unsigned curr = l - bits;
        long room = 1l << curr;
        while (curr + bits < max_len)
        {
          room -= static_cast<long>(rem[curr + bits]);
          if (room <= 0)
            break;
          ++curr;
          room <<= 1;
        }
        sub_bits = curr;
        if (next_sub + (1u << sub_bits) > cap)
          return false;
        sub_prefix = prefix;
        sub_at = next_sub;
        next_sub += 1u << sub_bits;
        for (unsigned i = 0; i < (1u << sub_bits); ++i)
          table[sub_at + i] = entry(K_BAD, 0, 0, 1);
        table[prefix] = entry(K_SUB, sub_at, sub_bits, bits);
      }
      uint32_t const e = symbol_entry(litlen, sorted[k], l - bits);
      for (unsigned i = rev >> bits; i < (1u << sub_bits); i += 1u << (l - bits))
        table[sub_at + i] = e;
      --rem[l];
    }
  return true;
}

struct Tables
{
  uint32_t litlen[LITLEN_SIZE];
  uint32_t dist[DIST_SIZE];
};
} // namespace inflate_detail

// Inflates the raw DEFLATE stream in[0, in_len) into exactly out_len bytes at out.  8 bytes behind `in + in_len` must be readable.
inline bool inflate_raw(uint8_t const * in, size_t in_len, uint8_t * out, size_t out_len)
{
  using namespace inflate_detail;
  uint8_t const * ip = in;
  uint8_t const * const in_end = in + in_len;
  uint8_t * op = out;
  uint8_t * const out_end = out + out_len;
  uint64_t bitbuf = 0;
  unsigned bitcnt = 0;
  // refill to at least 56 bits (the bytes behind in_end read as whatever lies there -- at most 8 of them, never used for
  // output that counts: `overrun` tells when more bits were consumed than the stream has)
  auto refill = [&]()
  {
    if (ip <= in_end) // (reads in_end + 8 at most)
    {
      uint64_t w;
      std::memcpy(&w, ip, 8);
      bitbuf |= w << bitcnt;
    }
    ip += (63 - bitcnt) >> 3; // (behind the stream: zeros come in, and `overrun` says so before they count)
    bitcnt |= 56;
  };
  auto overrun = [&]() { return (ip - in_end) * 8 > static_cast<long>(bitcnt); }; // more bits taken than the stream has
  static thread_local Tables t;
  static thread_local bool fixed_ready = false;
  static thread_local Tables fixed;
  for (bool last = false; !last;)
  {
    if (overrun())
      return false;
    refill();
    last = bitbuf & 1u;
    unsigned const type = (bitbuf >> 1) & 3u;
    bitbuf >>= 3;
    bitcnt -= 3;
    Tables const * tab = &t;
    if (type == 0)
    {
      // stored: back to the byte boundary, LEN, NLEN, bytes
      unsigned const drop = bitcnt & 7u;
      bitbuf >>= drop;
      bitcnt -= drop;
      ip -= bitcnt >> 3; // the whole bytes still in the buffer go back
      bitbuf = 0;
      bitcnt = 0;
      if (ip + 4 > in_end)
        return false;
      unsigned const len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
      ip += 4;
      if ((len ^ nlen) != 0xFFFFu || len > static_cast<size_t>(in_end - ip) || len > static_cast<size_t>(out_end - op))
        return false;
      if (len)
        std::memcpy(op, ip, len);
      ip += len;
      op += len;
      continue;
    }
    if (type == 3)
      return false;
    if (type == 1)
    {
      if (!fixed_ready)
      {
        uint8_t lens[288];
        for (unsigned i = 0; i < 288; ++i)
          lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        uint8_t dl[32];
        std::memset(dl, 5, 32);
        if (!build_table(lens, 288, true, LITLEN_BITS, fixed.litlen, LITLEN_SIZE) || !build_table(dl, 32, false, DIST_BITS, fixed.dist, DIST_SIZE))
          return false;
        fixed_ready = true;
      }
      tab = &fixed;
    }
    else
    {
      // dynamic: HLIT, HDIST, HCLEN, the code-length code, then the lengths of both alphabets in one run
      unsigned const hlit = (bitbuf & 31u) + 257, hdist = ((bitbuf >> 5) & 31u) + 1, hclen = ((bitbuf >> 10) & 15u) + 4;
      bitbuf >>= 14;
      bitcnt -= 14;
      if (hlit > 286 || hdist > 30)
        return false;
      static uint8_t const ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (unsigned i = 0; i < hclen; ++i)
      {
        if (bitcnt < 3)
          refill();
        cl[ORDER[i]] = bitbuf & 7u;
        bitbuf >>= 3;
        bitcnt -= 3;
      }
      // (the code-length code: 7 bits at most -- a flat table of 128 entries: symbol << 3 | length, 0 = unused)
      uint8_t clt[128] = {0};
      {
        unsigned count[8] = {0};
        for (unsigned i = 0; i < 19; ++i)
          ++count[cl[i]];
        long left = 1;
        for (unsigned l = 1; l <= 7; ++l)
        {
          left = (left << 1) - static_cast<long>(count[l]);
          if (left < 0)
            return false;
        }
        if (left > 0) // (incomplete: zlib refuses that of this code always)
          return false;
        unsigned code = 0;
        for (unsigned l = 1; l <= 7; ++l, code <<= 1)
          for (unsigned s = 0; s < 19; ++s)
            if (cl[s] == l)
            {
              unsigned rev = 0;
              for (unsigned b = 0; b < l; ++b)
                rev |= ((code >> b) & 1u) << (l - 1 - b);
              for (unsigned i = rev; i < 128; i += 1u << l)
                clt[i] = static_cast<uint8_t>((s << 3) | l);
              ++code;
            }
      }
      uint8_t lens[286 + 30 + 138];
      unsigned const total = hlit + hdist;
      for (unsigned i = 0; i < total;)
      {
        if (overrun())
          return false;
        refill();
        unsigned const e = clt[bitbuf & 127u];
        if (e == 0)
          return false;
        bitbuf >>= e & 7u;
        bitcnt -= e & 7u;
        unsigned const sym = e >> 3;
        if (sym < 16)
        {
          lens[i++] = static_cast<uint8_t>(sym);
          continue;
        }
        unsigned rep, val = 0;
        if (sym == 16)
        {
          if (i == 0)
            return false;
          val = lens[i - 1];
          rep = 3 + (bitbuf & 3u);
          bitbuf >>= 2;
          bitcnt -= 2;
        }
        else if (sym == 17)
        {
          rep = 3 + (bitbuf & 7u);
          bitbuf >>= 3;
          bitcnt -= 3;
        }
        else
        {
          rep = 11 + (bitbuf & 127u);
          bitbuf >>= 7;
          bitcnt -= 7;
        }
        if (i + rep > total)
          return false;
        std::memset(lens + i, static_cast<int>(val), rep);
        i += rep;
      }
      if (lens[256] == 0) // no end-of-block code
        return false;
      if (!build_table(lens, hlit, true, LITLEN_BITS, t.litlen, LITLEN_SIZE) || !build_table(lens + hlit, hdist, false, DIST_BITS, t.dist, DIST_SIZE))
        return false;
    }
    uint32_t const * const lt = tab->litlen;
    uint32_t const * const dt = tab->dist;
    // ---- symbols of the block ----
    for (;;)
    {
      bool const fast = ip + 8 <= in_end && op + 258 + 8 <= out_end;
      if (!fast && overrun())
        return false;
      refill();
      uint32_t e = lt[bitbuf & ((1u << LITLEN_BITS) - 1)];
      if ((e & K_MASK) == K_SUB)
      {
        bitbuf >>= LITLEN_BITS;
        bitcnt -= LITLEN_BITS;
        e = lt[(e >> 16) + (bitbuf & ((1u << ((e >> 8) & 31u)) - 1))];
      }
      bitbuf >>= e & 31u;
      bitcnt -= e & 31u;
      if ((e & K_MASK) == K_LIT)
      {
        if (fast)
        {
          *op++ = static_cast<uint8_t>(e >> 16);
          // two more literals out of the same 56 bits, when they are literals of the first level (at most 3 x 11 + ... bits used)
          e = lt[bitbuf & ((1u << LITLEN_BITS) - 1)];
          if ((e & K_MASK) == K_LIT)
          {
            bitbuf >>= e & 31u;
            bitcnt -= e & 31u;
            *op++ = static_cast<uint8_t>(e >> 16);
            e = lt[bitbuf & ((1u << LITLEN_BITS) - 1)];
            if ((e & K_MASK) == K_LIT)
            {
              bitbuf >>= e & 31u;
              bitcnt -= e & 31u;
              *op++ = static_cast<uint8_t>(e >> 16);
            }
          }
          continue;
        }
        if (op >= out_end)
          return false;
        *op++ = static_cast<uint8_t>(e >> 16);
        continue;
      }
      if ((e & K_MASK) == K_EOB)
        break;
      if ((e & K_MASK) != K_LEN)
        return false;
      unsigned const lx = (e >> 8) & 31u;
      size_t const length = (e >> 16) + (bitbuf & ((1u << lx) - 1));
      bitbuf >>= lx;
      bitcnt -= lx;
      // (at most 15 + 5 bits are gone: 36 left, a distance takes 15 + 13)
      uint32_t d = dt[bitbuf & ((1u << DIST_BITS) - 1)];
      if ((d & K_MASK) == K_SUB)
      {
        bitbuf >>= DIST_BITS;
        bitcnt -= DIST_BITS;
        d = dt[(d >> 16) + (bitbuf & ((1u << ((d >> 8) & 31u)) - 1))];
      }
      bitbuf >>= d & 31u;
      bitcnt -= d & 31u;
      if ((d & K_MASK) != K_LEN)
        return false;
      unsigned const dx = (d >> 8) & 31u;
      size_t const dist = (d >> 16) + (bitbuf & ((1u << dx) - 1));
      bitbuf >>= dx;
      bitcnt -= dx;
      if (dist > static_cast<size_t>(op - out) || length > static_cast<size_t>(out_end - op))
        return false;
      uint8_t const * from = op - dist;
      if (fast && dist >= 8)
      {
        // 8 bytes at a time; may write up to 7 bytes behind the match (room is there: the fast condition)
        uint8_t * const stop = op + length;
        do
        {
          std::memcpy(op, from, 8);
          op += 8;
          from += 8;
        } while (op < stop);
        op = stop;
      }
      else if (dist == 1)
      {
        std::memset(op, *from, length);
        op += length;
      }
      else
        for (size_t i = 0; i < length; ++i)
          *op++ = *from++;
    }
    if (overrun())
      return false;
  }
  return op == out_end;
}

// CRC-32 of a member's data (the gzip polynomial, reflected), eight bytes a step over eight tables: a BGZF member carries it
// behind its deflate stream, and a reader that inflates with its own decoder owes the file that comparison.
struct Crc32Tables
{
  uint32_t t[8][256];
  Crc32Tables()
  {
    for (uint32_t i = 0; i < 256; ++i)
    {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k)
        c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k)
        t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 255u];
  }
};

inline uint32_t crc32_of(uint8_t const * p, size_t n)
{
  static Crc32Tables const T;
  uint32_t c = 0xFFFFFFFFu;
  for (; n >= 8; n -= 8, p += 8)
  {
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 255u] ^ T.t[6][(w >> 8) & 255u] ^ T.t[5][(w >> 16) & 255u] ^ T.t[4][(w >> 24) & 255u] ^ T.t[3][(w >> 32) & 255u] ^
        T.t[2][(w >> 40) & 255u] ^ T.t[1][(w >> 48) & 255u] ^ T.t[0][w >> 56];
  }
  for (; n; --n, ++p)
    c = T.t[0][(c ^ *p) & 255u] ^ (c >> 8);
  return ~c;
}
} // namespace gtx
