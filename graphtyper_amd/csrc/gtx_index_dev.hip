// gtx_index_dev.hip -- the k-mer index built ON THE DEVICE (gfx950): from the emitted (key, label) pairs of the sweep to
// every table the kernels read -- grouping by key, the bucketed exact table, the two half-key tables, and the tables of
// the position-hinted pass.  Replaces, for contexts with a device, the host stages of gtx_host.cpp: build_tables_host,
// which took 0.2-0.3 s per 1 Mb region (30x the device time of the region's reads); the host only enumerates the 32-mers
// (index_graph's sweep, indexer.cpp:246-291) and derives the small per-position arrays of the graph.
//
// Stages (every one a data-parallel kernel or a rocPRIM primitive over ~1 M elements per Mb):
//   stable radix sort of the keys            bucket order inside a key = emission order, as PHIndex holds it
//   heads + scan                              unique keys, key_off
//   labels in key order                       DevLabel with (site, allele) resolved
//   groups by first / last 16 bases           the half-key buckets (the last-16 order is a second stable sort)
//   hash tables                               slots claimed with atomicCAS on the count word, filled front to back
//   hint tables                               one thread per key (neighbour verdict, filters), one per position (flags)
// The per-key / per-position decisions are the text of index_build.hpp, shared with the host build.
#include <cstring> // (rocPRIM's texture iterator calls memset on the host)

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "gtx_ctx.hpp"
#include "gtx_devmem.hpp"
#include "index_build.hpp"

namespace gtx
{
namespace
{
bool ok_hip(hipError_t e, char const * what)
{
  if (e == hipSuccess)
    return true;
  g_last_error = std::string("index build: ") + what + ": " + hipGetErrorString(e);
  return false;
}

// device buffers of the build; the ones the kernels keep reading are handed to the context, the rest is freed
struct Pool
{
  std::vector<void *> temps;
  bool fine = true;
  template <class T>
  T * get(size_t n, char const * what, bool zero = false)
  {
    void * p = nullptr;
    if (!fine)
      return nullptr;
    fine = ok_hip(gtx::dev_malloc(&p, (n ? n : 1) * sizeof(T)), what);
    if (fine && zero)
      fine = ok_hip(gtx::dev_zero_async(p, (n ? n : 1) * sizeof(T)), what); // (its first user is a later launch on the same stream)
    if (p)
      temps.push_back(p);
    return static_cast<T *>(p);
  }
  template <class T>
  T * keep(T * p, std::vector<void *> & owner) // ownership moves to the context
  {
    auto it = std::find(temps.begin(), temps.end(), static_cast<void *>(p));
    if (it != temps.end())
    {
      temps.erase(it);
      owner.push_back(p);
    }
    return p;
  }
  ~Pool()
  {
    if (!temps.empty())
      (void)hipStreamSynchronize(gtx::tls_build_stream); // (on an error path kernels may still be writing to them; freed blocks are handed out again)
    for (void * p : temps)
      (void)gtx::dev_free(p);
  }
};

constexpr uint32_t TB = 256;
inline uint32_t blocks_for(uint64_t n) { return static_cast<uint32_t>((n + TB - 1) / TB); }

__global__ void k_iota(uint32_t * v, uint32_t n)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    v[i] = i;
}

__global__ void k_heads(uint64_t const * keys, uint32_t n, uint32_t shift, uint64_t mask, uint32_t * head)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    head[i] = (i == 0 || ((keys[i] >> shift) & mask) != ((keys[i - 1] >> shift) & mask)) ? 1u : 0u;
}

__global__ void k_heads32(uint32_t const * keys, uint32_t n, uint32_t * head)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// unique keys and their label ranges from the heads of the sorted emission list
__global__ void k_unique(uint64_t const * sorted, uint32_t const * head, uint32_t const * kidx, uint32_t n, uint64_t * keys, uint32_t * key_off,
                         uint32_t n_keys)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i])
  {
    keys[kidx[i]] = sorted[i];
    key_off[kidx[i]] = i;
  }
  if (i == 0)
    key_off[n_keys] = n;
}

__global__ void k_labels(GraphView g, gtx_label const * in, uint32_t const * perm, uint32_t n, gtx_label * sorted, DevLabel * dev)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  gtx_label const l = in[perm[i]];
  sorted[i] = l;
  DevLabel d{l.start_index, l.end_index, INVALID, 0};
  if (l.variant_id != INVALID)
  {
    d.site = g.var_out_ref[l.variant_id] - 1;
    d.allele = l.variant_id - g.ref_first_var[d.site];
  }
  dev[i] = d;
}

// plane-form key (gtx_flat.hpp: plane_key) on the device
__device__ inline uint64_t dev_plane_key(uint64_t key)
{
  auto compact_even_bits = [](uint64_t x)
  {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return static_cast<uint32_t>(x);
  };
  uint64_t const lo = __brev(compact_even_bits(key)), hi = __brev(compact_even_bits(key >> 1));
  return (hi << 32) | lo;
}

__device__ inline uint64_t dev_half_of(uint64_t pk, int side)
{
  uint32_t const lo = static_cast<uint32_t>(pk), hi = static_cast<uint32_t>(pk >> 32);
  return side == 0 ? ((lo & 0xFFFFu) | ((hi & 0xFFFFu) << 16)) : ((lo >> 16) | (hi & 0xFFFF0000u));
}

// claims a slot of a bucketed table (gtx_flat.hpp: BUCKET_SLOTS) for `s`: the count word goes from 0 to s.cnt with one
// compare-and-swap, slots of a bucket are tried front to back (a lookup stops at a bucket whose last slot is empty), a full
// bucket spills into the next one; the other fields are written behind the claim (nobody reads before the kernel ends)
__device__ inline void table_insert(IndexSlot * slots, uint32_t log2_buckets, IndexSlot const & s)
{
  uint64_t const mask = (1ull << log2_buckets) - 1;
  for (uint64_t b = hash_key(s.key, log2_buckets);; b = (b + 1) & mask)
    for (uint32_t k = 0; k < BUCKET_SLOTS; ++k)
    {
      IndexSlot * slot = slots + b * BUCKET_SLOTS + k;
      if (atomicCAS(&slot->cnt, 0u, s.cnt) == 0u)
      {
        slot->key = s.key;
        slot->off = s.off;
        slot->p[0] = s.p[0];
        slot->p[1] = s.p[1];
        slot->p[2] = s.p[2];
        slot->p[3] = s.p[3];
        return;
      }
    }
}

__global__ void k_exact_table(uint64_t const * keys, uint32_t const * key_off, DevLabel const * labels, uint32_t n_keys, uint64_t * pk,
                              IndexSlot * slots, uint32_t log2_cap, uint32_t * several)
{
  uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_keys)
    return;
  uint64_t const key = dev_plane_key(keys[k]);
  pk[k] = key;
  IndexSlot s{key, key_off[k], key_off[k + 1] - key_off[k], {0, 0, 0, 0}};
  if (s.cnt == 1) // inline copy of the one label
  {
    DevLabel const d = labels[s.off];
    s.p[0] = d.start;
    s.p[1] = d.end;
    s.p[2] = d.site;
    s.p[3] = d.allele;
  }
  else
    atomicAdd(several, 1u);
  table_insert(slots, log2_cap, s);
}

// group bookkeeping: begin index of every element's group from the heads (a running maximum), sizes by counting
__global__ void k_group_begin_seed(uint32_t const * head, uint32_t n, uint32_t * seed)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    seed[i] = head[i] ? i : 0u;
}

__global__ void k_group_count(uint32_t const * begin, uint32_t n, uint32_t * count)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    atomicAdd(count + begin[i], 1u);
}

__global__ void k_left_groups(uint32_t const * begin, uint32_t const * count, uint32_t n, uint32_t * lsize)
{
  uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n)
    lsize[k] = count[begin[k]];
}

__global__ void k_right_groups(uint32_t const * rorder, uint32_t const * gbegin, uint32_t const * count, uint32_t n, uint32_t * rbegin, uint32_t * rsize)
{
  uint32_t const m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < n)
  {
    rbegin[rorder[m]] = gbegin[m];
    rsize[rorder[m]] = count[gbegin[m]];
  }
}

__global__ void k_low_keys(uint64_t const * keys, uint32_t n, uint32_t * low)
{
  uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n)
    low[k] = static_cast<uint32_t>(keys[k]);
}

// hlist = every key twice, grouped by its first 16 bases (key order) and by its last 16 bases (rorder), and one slot per
// group in hslots (a single entry inline)
__global__ void k_half_tables(uint64_t const * pk, uint32_t const * key_off, uint32_t const * lbegin, uint32_t const * lsize, uint32_t const * rorder,
                              uint32_t const * rhead, uint32_t const * rgbegin, uint32_t const * rcount, uint32_t n, HalfEntry * hlist,
                              IndexSlot * hslots, uint32_t h_log2_cap)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  // side 0: element i of the key order
  {
    HalfEntry const he{pk[i], key_off[i], key_off[i + 1] - key_off[i]};
    hlist[i] = he;
    if (lbegin[i] == i)
    {
      IndexSlot s{dev_half_of(he.key, 0), i, lsize[i], {0, 0, 0, 0}};
      if (s.cnt == 1)
      {
        s.p[0] = static_cast<uint32_t>(he.key);
        s.p[1] = static_cast<uint32_t>(he.key >> 32);
        s.p[2] = he.off;
        s.p[3] = he.cnt;
      }
      table_insert(hslots, h_log2_cap, s);
    }
  }
  // side 1: element i of the order by the last 16 bases
  {
    uint32_t const k = rorder[i];
    HalfEntry const he{pk[k], key_off[k], key_off[k + 1] - key_off[k]};
    hlist[n + i] = he;
    if (rhead[i])
    {
      IndexSlot s{dev_half_of(he.key, 1) | (1ull << 32), n + i, rcount[rgbegin[i]], {0, 0, 0, 0}};
      if (s.cnt == 1)
      {
        s.p[0] = static_cast<uint32_t>(he.key);
        s.p[1] = static_cast<uint32_t>(he.key >> 32);
        s.p[2] = he.off;
        s.p[3] = he.cnt;
      }
      table_insert(hslots, h_log2_cap, s);
    }
  }
}

__global__ void k_judge_keys(HintKeys t, uint32_t * nb, uint8_t * nb_same, uint32_t * filt0, uint32_t * filt1, uint32_t filt_log2,
                             uint64_t const * pk, IndexSlot * slots, uint32_t log2_cap)
{
  uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t.n_keys)
    return;
  uint32_t n = 0, same = 1, known = 0;
  hint_judge_key(t, k, n, same, known);
  nb[k] = n;
  nb_same[k] = static_cast<uint8_t>(same | (known << 1));
  if (known && n != 0 && slots)
  {
    // the key's slot of the exact table gets SLOT_NB_KNOWN (gtx_flat.hpp)
    uint64_t const key = pk[k], mask = (1ull << log2_cap) - 1;
    bool done = false;
    for (uint64_t b = hash_key(key, log2_cap); !done; b = (b + 1) & mask)
      for (uint32_t j = 0; j < BUCKET_SLOTS && !done; ++j)
      {
        IndexSlot * sl = slots + b * BUCKET_SLOTS + j;
        if (sl->cnt == 0)
          done = true; // (cannot happen: every key is in the table)
        else if (sl->key == key)
        {
          atomicOr(&sl->off, SLOT_NB_KNOWN);
          done = true;
        }
      }
  }
  for (uint32_t side = 0; side < 2; ++side)
  {
    uint32_t w0, w1, word, mask;
    hint_half_planes(static_cast<uint32_t>(side == 0 ? t.keys[k] >> 32 : t.keys[k]), w0, w1);
    hint_filter_slot(w0, w1, filt_log2, word, mask);
    atomicOr((side == 0 ? filt0 : filt1) + word, mask);
  }
}

__global__ void k_position_flags(GraphView g, HintKeys t, uint32_t const * nb, uint8_t const * nb_same, uint8_t const * base, uint8_t const * room,
                                 uint8_t const * back, uint32_t n, uint2_t * flags)
{
  uint32_t const p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n)
    flags[p] = hint_position_flags(g, t, nb, nb_same, base, room, back, n, p);
}

// ---- the sweep's k-mers that lie inside one reference node of plain A/C/G/T (gtx_flat.hpp: EmitRun), one thread each:
// the k-mer that starts at base k of the run's node, at its place in the sweep's order (host_before listed k-mers and
// dev_before run k-mers in front of the run: index host_before + t for the t-th run k-mer overall)
__global__ void k_emit_runs(GraphView g, EmitRun const * __restrict__ runs, uint32_t n_runs, uint32_t total, uint64_t * __restrict__ keys,
                            gtx_label * __restrict__ labels)
{
  uint32_t const t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total)
    return;
  uint32_t lo = 0, hi = n_runs; // the last run with dev_before <= t
  while (hi - lo > 1)
  {
    uint32_t const mid = (lo + hi) >> 1;
    if (runs[mid].dev_before <= t)
      lo = mid;
    else
      hi = mid;
  }
  EmitRun const r = runs[lo];
  uint32_t const k = t - r.dev_before;
  char const * dna = g.dna + g.ref_dna[r.node] + k;
  uint64_t key = 0;
#pragma unroll 8
  for (uint32_t i = 0; i < K; ++i) // (codes 1 2 4 8 = A C G T: the host checked the node)
    key = (key << 2) | static_cast<uint64_t>(__ffs(static_cast<int>(dna[i])) - 1);
  uint32_t const at = r.host_before + t, first = g.ref_order[r.node] + k;
  keys[at] = key;
  labels[at] = gtx_label{first, first + (K - 1), INVALID};
}

// ... and the listed ones to theirs: the j-th listed k-mer has the k-mers of every run with host_before <= j in front of it
__global__ void k_place_listed(EmitRun const * __restrict__ runs, uint32_t n_runs, uint64_t const * __restrict__ lkeys,
                               gtx_label const * __restrict__ llabels, uint32_t n, uint64_t * __restrict__ keys, gtx_label * __restrict__ labels)
{
  uint32_t const j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n)
    return;
  uint32_t lo = 0, hi = n_runs; // number of runs with host_before <= j
  while (lo < hi)
  {
    uint32_t const mid = (lo + hi) >> 1;
    if (runs[mid].host_before <= j)
      lo = mid + 1;
    else
      hi = mid;
  }
  uint32_t const at = j + (lo ? runs[lo - 1].dev_before + runs[lo - 1].count : 0u);
  keys[at] = lkeys[j];
  labels[at] = llabels[j];
}

// ---- what the position-hinted pass needs of the graph alone (gtx_host.cpp: hint_graph_tables, the host's form): per
// position of the linear reference its nibble, the bases of its reference node in front of and behind it, the site behind
// its node as a tail walk may cross it; the reference as bit planes (a wavefront's 64 positions are two plane groups)
__global__ void k_hint_graph(GraphView g, uint32_t n, uint8_t * __restrict__ base, uint8_t * __restrict__ room, uint8_t * __restrict__ back,
                             uint2_t * __restrict__ tail, uint32_t * __restrict__ refp)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t b = 0; // (behind the last position: no plane bit)
  if (i < n)
  {
    auto nib = [](char c) -> uint32_t { return (c == 1 || c == 2 || c == 4 || c == 8) ? static_cast<uint32_t>(c) : 15u; };
    uint32_t const order = g.first_order + i;
    uint32_t lo = 0, hi = g.n_ref; // the last reference node that starts at or in front of the position
    while (hi - lo > 1)
    {
      uint32_t const mid = (lo + hi) >> 1;
      if (g.ref_order[mid] <= order)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t const r = lo, d = order - g.ref_order[r], len = g.ref_len[r];
    uint32_t rm = 0, bk = 0;
    uint2_t ti{0, 0};
    b = 15;
    if (d < len)
    {
      b = hint_plane_code(static_cast<uint8_t>(g.dna[g.ref_dna[r] + d]));
      rm = len - d < 255 ? len - d : 255;
      bk = d < 255 ? d : 255;
      if (r + 1 < g.n_ref && !g.is_sv_graph)
      {
        uint32_t const fv = g.ref_first_var[r], nv = g.ref_nvar[r];
        bool snp = nv >= 2 && nv <= 4;
        uint32_t codes = 0;
        ti = uint2_t{HINT_TAIL_NODE, r};
        for (uint32_t a = 0; a < nv && snp; ++a)
        {
          uint32_t const c = g.var_len[fv + a] == 1 ? nib(g.dna[g.var_dna[fv + a]]) : 15u;
          snp = c != 15;
          codes |= c << (4 * a);
        }
        if (snp)
        {
          uint32_t const next_len = g.ref_len[r + 1] < 255 ? g.ref_len[r + 1] : 255;
          ti = uint2_t{HINT_TAIL_NODE | HINT_TAIL_OK | (nv << HINT_TAIL_NALL_SHIFT) | (next_len << HINT_TAIL_NEXT_SHIFT) | (codes << HINT_TAIL_CODES_SHIFT), r};
        }
      }
    }
    else if (r + 1 < g.n_ref && g.ref_nvar[r] != 0)
    {
      uint32_t const v = g.ref_first_var[r], dv = order - g.var_order[v]; // (allele 0 of the site: the linear reference)
      if (order >= g.var_order[v] && dv < g.var_len[v])
        b = hint_plane_code(static_cast<uint8_t>(g.dna[g.var_dna[v] + dv]));
    }
    base[i] = static_cast<uint8_t>(b);
    room[i] = static_cast<uint8_t>(rm);
    back[i] = static_cast<uint8_t>(bk);
    tail[i] = ti;
  }
#pragma unroll
  for (uint32_t bit = 0; bit < 4; ++bit)
  {
    unsigned long long const m = __ballot((b >> bit) & 1u);
    if ((threadIdx.x & 63u) == 0)
    {
      refp[4 * (i >> 5) + bit] = static_cast<uint32_t>(m);
      refp[4 * ((i >> 5) + 1) + bit] = static_cast<uint32_t>(m >> 32);
    }
  }
}

// ---- the allele windows behind the linear reference (IndexView::win; index_build.hpp): a thread per window position.  The
// windows start at a multiple of 64 positions and are a multiple of 64 long: a wavefront's positions are two plane groups.
__global__ void k_window_cells(GraphView g, HintWindow const * __restrict__ win, uint32_t n_win, uint32_t win_base, uint32_t n_main, uint8_t * base,
                               uint8_t * room, uint8_t * back, uint2_t * tail, uint32_t * __restrict__ refp)
{
  uint32_t const t = blockIdx.x * blockDim.x + threadIdx.x, w = t / HINT_WIN_STRIDE, local = t % HINT_WIN_STRIDE;
  uint32_t b = 0;
  uint32_t const p = win_base + t;
  if (w < n_win)
  {
    uint8_t c = 15, rm = 0, bk = 0;
    uint2_t ti{0, 0};
    hint_window_cell(g, win[w], local, base, room, back, tail, n_main, c, rm, bk, ti);
    base[p] = c;
    room[p] = rm;
    back[p] = bk;
    tail[p] = ti;
    b = c;
  }
#pragma unroll
  for (uint32_t bit = 0; bit < 4; ++bit)
  {
    unsigned long long const m = __ballot((b >> bit) & 1u);
    if ((threadIdx.x & 63u) == 0 && w < n_win)
    {
      refp[4 * (p >> 5) + bit] = static_cast<uint32_t>(m);
      refp[4 * ((p >> 5) + 1) + bit] = static_cast<uint32_t>(m >> 32);
    }
  }
}

__global__ void k_window_flags(GraphView g, HintKeys t, uint32_t const * nb, uint8_t const * nb_same, uint8_t const * base, uint8_t const * room,
                               uint8_t const * back, uint32_t n_total, uint32_t n_main, HintWindow const * __restrict__ win, uint32_t n_win, uint32_t win_base,
                               uint2_t * flags)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x, w = i / HINT_WIN_STRIDE, local = i % HINT_WIN_STRIDE;
  if (w < n_win)
    flags[win_base + i] = hint_window_flags(g, t, nb, nb_same, base, room, back, n_total, flags, n_main, win[w], local, win_base + i);
}

template <class T>
bool to_device(Pool & pool, T *& d, std::vector<T> const & h, char const * what)
{
  d = pool.get<T>(h.size(), what);
  // (asynchronous: `h` has to live until the stream has been waited for -- build_index_device's host vectors are declared in front of its pool)
  return pool.fine && (h.empty() || ok_hip(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, gtx::tls_build_stream), what));
}

// rocPRIM primitives with their temporary storage
template <class K, class V>
bool sort_pairs(Pool & pool, K const * kin, K * kout, V const * vin, V * vout, uint32_t n, unsigned bits)
{
  size_t bytes = 0;
  if (!ok_hip(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, bits), "radix sort (size)"))
    return false;
  void * tmp = pool.get<uint8_t>(bytes, "radix sort storage");
  return pool.fine && ok_hip(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, gtx::tls_build_stream), "radix sort");
}

bool exclusive_sum(Pool & pool, uint32_t const * in, uint32_t * out, uint32_t n)
{
  size_t bytes = 0;
  if (!ok_hip(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>()), "scan (size)"))
    return false;
  void * tmp = pool.get<uint8_t>(bytes, "scan storage");
  return pool.fine && ok_hip(rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), gtx::tls_build_stream), "scan");
}

bool running_max(Pool & pool, uint32_t const * in, uint32_t * out, uint32_t n)
{
  size_t bytes = 0;
  if (!ok_hip(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::maximum<uint32_t>()), "max scan (size)"))
    return false;
  void * tmp = pool.get<uint8_t>(bytes, "max scan storage");
  return pool.fine && ok_hip(rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<uint32_t>(), gtx::tls_build_stream), "max scan");
}

// begin index and size of every element's group, from its head flags
bool groups_of(Pool & pool, uint32_t const * head, uint32_t n, uint32_t *& begin, uint32_t *& count)
{
  uint32_t * seed = pool.get<uint32_t>(n, "group seeds");
  begin = pool.get<uint32_t>(n, "group begins");
  count = pool.get<uint32_t>(n, "group counts", true);
  if (!pool.fine)
    return false;
  hipLaunchKernelGGL(k_group_begin_seed, dim3(blocks_for(n)), dim3(TB), 0, gtx::tls_build_stream, head, n, seed);
  if (!running_max(pool, seed, begin, n))
    return false;
  hipLaunchKernelGGL(k_group_count, dim3(blocks_for(n)), dim3(TB), 0, gtx::tls_build_stream, begin, n, count);
  return ok_hip(hipGetLastError(), "group kernels");
}
} // namespace

// Builds every index table on the device of `c` (its graph is already there: c.dev_graph) from the sweep's emission list.
// Fills c.dev_index and the index facts the host keeps (key / label counts, the pass-1 build choice, device copies of
// keys / key_off / labels for the inspection entry points).
int build_index_device(gtx_ctx & c, std::vector<Emit> const & em, std::vector<EmitRun> const & runs)
{
  // (host sources of asynchronous copies: declared in front of the pool, whose destructor waits for the stream)
  std::vector<uint64_t> h_keys;
  std::vector<gtx_label> h_labels;
  std::vector<HintWindow> win;
  std::vector<uint32_t> site_win;
  hipStream_t const bs = gtx::tls_build_stream;
  Pool pool;
  uint32_t const n_listed = static_cast<uint32_t>(em.size());
  uint64_t const n_run_kmers = runs.empty() ? 0ull : static_cast<uint64_t>(runs.back().dev_before) + runs.back().count;
  if (em.size() + n_run_kmers >= (1ull << 31))
  {
    g_last_error = "index build: more than 2^31 indexed k-mers in one region";
    return GTX_ERR_UNSUPPORTED;
  }
  uint32_t const E = n_listed + static_cast<uint32_t>(n_run_kmers);
  IndexView ix{};
  ix.max_index_labels = static_cast<uint32_t>(c.params.max_index_labels);
  ix.half_bucket_cap = HALF_BUCKET_CAP;
  if (char const * e = std::getenv("GTX_HALF_BUCKET_CAP")) // A/B switch for benchmarking: 0 = probe the 96 neighbours directly
    ix.half_bucket_cap = static_cast<uint32_t>(std::min<long>(std::max<long>(std::atol(e), 0), HALF_BUCKET_CAP));
  // ---- emission list on the device (structure of arrays): the listed k-mers from the host, the runs' made here
  uint64_t *d_keys_in = pool.get<uint64_t>(E, "emitted keys"), *d_sorted = pool.get<uint64_t>(E, "sorted keys");
  gtx_label * d_labels_in = pool.get<gtx_label>(E, "emitted labels");
  uint32_t *d_iota = pool.get<uint32_t>(E, "iota"), *d_perm = pool.get<uint32_t>(E, "permutation");
  if (!pool.fine)
    return GTX_ERR_HIP;
  if (runs.empty())
  {
    h_keys.resize(E);
    h_labels.resize(E);
    for (uint32_t i = 0; i < E; ++i)
    {
      h_keys[i] = em[i].key;
      h_labels[i] = em[i].label;
    }
    if (E != 0 && (!ok_hip(hipMemcpyAsync(d_keys_in, h_keys.data(), E * sizeof(uint64_t), hipMemcpyHostToDevice, bs), "emitted keys") ||
                   !ok_hip(hipMemcpyAsync(d_labels_in, h_labels.data(), E * sizeof(gtx_label), hipMemcpyHostToDevice, bs), "emitted labels")))
      return GTX_ERR_HIP;
  }
  else
  {
    h_keys.resize(n_listed);
    h_labels.resize(n_listed);
    for (uint32_t i = 0; i < n_listed; ++i)
    {
      h_keys[i] = em[i].key;
      h_labels[i] = em[i].label;
    }
    uint64_t * d_lkeys = nullptr;
    gtx_label * d_llabels = nullptr;
    EmitRun * d_runs = nullptr;
    if (!to_device(pool, d_lkeys, h_keys, "listed keys") || !to_device(pool, d_llabels, h_labels, "listed labels") ||
        !to_device(pool, d_runs, runs, "k-mer runs"))
      return GTX_ERR_HIP;
    uint32_t const n_runs = static_cast<uint32_t>(runs.size());
    hipLaunchKernelGGL(k_emit_runs, dim3(blocks_for(n_run_kmers)), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, d_runs, n_runs,
                       static_cast<uint32_t>(n_run_kmers), d_keys_in, d_labels_in);
    if (n_listed)
      hipLaunchKernelGGL(k_place_listed, dim3(blocks_for(n_listed)), dim3(TB), 0, gtx::tls_build_stream, d_runs, n_runs, d_lkeys, d_llabels, n_listed, d_keys_in,
                         d_labels_in);
  }
  uint32_t n_keys = 0;
  uint64_t * d_keys = nullptr;
  uint32_t * d_key_off = nullptr;
  gtx_label * d_labels_sorted = pool.get<gtx_label>(E, "labels in key order");
  DevLabel * d_dev_labels = pool.get<DevLabel>(E, "device labels");
  if (E != 0)
  {
    // ---- group by key, keeping the emission order inside a key
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(E)), dim3(TB), 0, gtx::tls_build_stream, d_iota, E);
    if (!sort_pairs(pool, d_keys_in, d_sorted, d_iota, d_perm, E, 64))
      return GTX_ERR_HIP;
    uint32_t *d_head = pool.get<uint32_t>(E, "key heads"), *d_kidx = pool.get<uint32_t>(E, "key numbers");
    if (!pool.fine)
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_heads, dim3(blocks_for(E)), dim3(TB), 0, gtx::tls_build_stream, d_sorted, E, 0u, ~0ull, d_head);
    if (!exclusive_sum(pool, d_head, d_kidx, E))
      return GTX_ERR_HIP;
    uint32_t last_idx = 0, last_head = 0;
    if (!ok_hip(hipMemcpyAsync(&last_idx, d_kidx + (E - 1), 4, hipMemcpyDeviceToHost, bs), "key count") ||
        !ok_hip(hipMemcpyAsync(&last_head, d_head + (E - 1), 4, hipMemcpyDeviceToHost, bs), "key count") || !ok_hip(hipStreamSynchronize(bs), "key count"))
      return GTX_ERR_HIP;
    n_keys = last_idx + last_head;
    d_keys = pool.get<uint64_t>(n_keys, "keys");
    d_key_off = pool.get<uint32_t>(n_keys + 1, "key offsets");
    if (!pool.fine)
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_unique, dim3(blocks_for(E)), dim3(TB), 0, gtx::tls_build_stream, d_sorted, d_head, d_kidx, E, d_keys, d_key_off, n_keys);
    hipLaunchKernelGGL(k_labels, dim3(blocks_for(E)), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, d_labels_in, d_perm, E, d_labels_sorted, d_dev_labels);
  }
  else
  {
    d_keys = pool.get<uint64_t>(1, "keys");
    d_key_off = pool.get<uint32_t>(1, "key offsets", true);
  }
  // ---- exact table
  uint32_t log2_cap = 2;
  while ((static_cast<uint64_t>(BUCKET_SLOTS) << log2_cap) < 2ull * n_keys + 1)
    ++log2_cap;
  IndexSlot * d_slots = pool.get<IndexSlot>(static_cast<uint64_t>(BUCKET_SLOTS) << log2_cap, "index slots", true);
  uint64_t * d_pk = pool.get<uint64_t>(n_keys, "plane keys");
  uint32_t * d_several = pool.get<uint32_t>(1, "several-label counter", true);
  if (!pool.fine)
    return GTX_ERR_HIP;
  if (n_keys)
    hipLaunchKernelGGL(k_exact_table, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_keys, d_key_off, d_dev_labels, n_keys, d_pk, d_slots, log2_cap,
                       d_several);
  // ---- groups by the first 16 bases (key order) and by the last 16 bases (a second stable sort)
  uint32_t hl = 2;
  while ((static_cast<uint64_t>(BUCKET_SLOTS) << hl) < 4ull * n_keys + 1)
    ++hl;
  IndexSlot * d_hslots = pool.get<IndexSlot>(static_cast<uint64_t>(BUCKET_SLOTS) << hl, "half-key slots", true);
  HalfEntry * d_hlist = pool.get<HalfEntry>(2ull * n_keys, "half-key buckets");
  uint32_t *d_lbegin = nullptr, *d_lcount = nullptr, *d_lsize = pool.get<uint32_t>(n_keys, "left sizes");
  uint32_t *d_rorder = pool.get<uint32_t>(n_keys, "right order"), *d_rbegin = pool.get<uint32_t>(n_keys, "right begins"),
           *d_rsize = pool.get<uint32_t>(n_keys, "right sizes");
  uint32_t *d_rhead = nullptr, *d_rgbegin = nullptr, *d_rcount = nullptr;
  if (!pool.fine)
    return GTX_ERR_HIP;
  if (n_keys)
  {
    uint32_t * d_lhead = pool.get<uint32_t>(n_keys, "left heads");
    if (!pool.fine)
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_heads, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_keys, n_keys, 32u, 0xFFFFFFFFull, d_lhead);
    if (!groups_of(pool, d_lhead, n_keys, d_lbegin, d_lcount))
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_left_groups, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_lbegin, d_lcount, n_keys, d_lsize);
    uint32_t *d_low = pool.get<uint32_t>(n_keys, "low halves"), *d_low_sorted = pool.get<uint32_t>(n_keys, "low halves sorted"),
             *d_kiota = pool.get<uint32_t>(n_keys, "key iota");
    d_rhead = pool.get<uint32_t>(n_keys, "right heads");
    if (!pool.fine)
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_low_keys, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_keys, n_keys, d_low);
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_kiota, n_keys);
    if (!sort_pairs(pool, d_low, d_low_sorted, d_kiota, d_rorder, n_keys, 32))
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_heads32, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_low_sorted, n_keys, d_rhead);
    if (!groups_of(pool, d_rhead, n_keys, d_rgbegin, d_rcount))
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_right_groups, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_rorder, d_rgbegin, d_rcount, n_keys, d_rbegin, d_rsize);
    hipLaunchKernelGGL(k_half_tables, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, d_pk, d_key_off, d_lbegin, d_lsize, d_rorder, d_rhead, d_rgbegin,
                       d_rcount, n_keys, d_hlist, d_hslots, hl);
  }
  // ---- tables of the position-hinted pass
  uint32_t const fl = hint_filter_log2_words(static_cast<uint64_t>(n_keys));
  // (the graph's own part of them -- gtx_host.cpp: hint_graph_tables is the host's form -- is made here as well)
  uint32_t const R = static_cast<uint32_t>(c.graph.ref_order.size());
  uint32_t const hint_first = R ? c.graph.ref_order[0] - 1 : 0; // order = 1-based contig position
  uint32_t const hint_n = (R == 0 || R - 1 >= HINT_NO_SITE) ? 0u : c.graph.ref_order[R - 1] + c.graph.ref_len[R - 1] - c.graph.ref_order[0];
  bool const hints = hint_n != 0;
  uint32_t *d_f0 = pool.get<uint32_t>(hints ? (1ull << fl) : 1, "filter 0", true), *d_f1 = pool.get<uint32_t>(hints ? (1ull << fl) : 1, "filter 1", true);
  // (the allele windows continue the per-position tables behind win_base: gtx_flat.hpp)
  if (hints)
    hint_list_windows(c.graph, win, site_win);
  uint32_t const n_win = static_cast<uint32_t>(win.size()), win_base = hint_win_base(hint_n);
  uint64_t const hint_total = hint_total_positions(hint_n, n_win);
  uint2_t * d_flags = pool.get<uint2_t>(hints ? hint_total : 1, "position flags", true);
  // (padded: the kernel loads 6 plane groups from any position without a bounds test)
  uint32_t * d_refp = pool.get<uint32_t>(hints ? 4 * (static_cast<size_t>(hint_total) / 32 + 8) : 32, "reference planes", true);
  uint2_t * d_tail = pool.get<uint2_t>(hints ? hint_total : 1, "tail sites", !hints || n_win != 0);
  HintWindow * d_win = nullptr;
  uint32_t * d_site_win = nullptr;
  if (n_win && (!to_device(pool, d_win, win, "allele windows") || !to_device(pool, d_site_win, site_win, "windows of the sites")))
    return GTX_ERR_HIP;
  if (!pool.fine)
    return GTX_ERR_HIP;
  if (hints)
  {
    uint8_t *d_base = pool.get<uint8_t>(hint_total, "reference bases", n_win != 0), *d_room = pool.get<uint8_t>(hint_total, "node room", n_win != 0),
            *d_back = pool.get<uint8_t>(hint_total, "node back", n_win != 0);
    uint32_t * d_nb = pool.get<uint32_t>(n_keys, "neighbour labels");
    uint8_t * d_same = pool.get<uint8_t>(n_keys, "neighbour verdicts");
    if (!pool.fine)
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(k_hint_graph, dim3((hint_n + TB - 1) / TB), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, hint_n, d_base, d_room, d_back, d_tail, d_refp);
    HintKeys const t{d_keys, d_key_off, d_dev_labels, n_keys, d_lbegin, d_lsize, d_rorder, d_rbegin, d_rsize};
    if (n_keys)
    {
      char const * nbk = std::getenv("GTX_NB_KNOWN"); // test switch: 0 = no slot gets SLOT_NB_KNOWN
      hipLaunchKernelGGL(k_judge_keys, dim3(blocks_for(n_keys)), dim3(TB), 0, gtx::tls_build_stream, t, d_nb, d_same, d_f0, d_f1, fl, d_pk,
                         (nbk && nbk[0] == '0') ? static_cast<IndexSlot *>(nullptr) : d_slots, log2_cap);
    }
    // (the flags look at the filters k_judge_keys has just filled, in stream order: HINT_NEAR_FREE)
    HintKeys tf = t;
    tf.filt0 = d_f0;
    tf.filt1 = d_f1;
    tf.filt_log2 = fl;
    hipLaunchKernelGGL(k_position_flags, dim3(blocks_for(hint_n)), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, tf, d_nb, d_same, d_base, d_room, d_back, hint_n, d_flags);
    if (n_win)
    {
      uint32_t const cells = n_win * HINT_WIN_STRIDE;
      hipLaunchKernelGGL(k_window_cells, dim3((cells + TB - 1) / TB), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, d_win, n_win, win_base, hint_n, d_base, d_room, d_back,
                         d_tail, d_refp);
      hipLaunchKernelGGL(k_window_flags, dim3((cells + TB - 1) / TB), dim3(TB), 0, gtx::tls_build_stream, c.dev_graph, tf, d_nb, d_same, d_base, d_room, d_back,
                         static_cast<uint32_t>(hint_total), hint_n, d_win, n_win, win_base, d_flags);
    }
  }
  uint32_t several = 0;
  if (!ok_hip(hipGetLastError(), "kernels") || !ok_hip(hipMemcpyAsync(&several, d_several, 4, hipMemcpyDeviceToHost, bs), "several-label counter") ||
      !ok_hip(hipStreamSynchronize(bs), "kernels"))
    return GTX_ERR_HIP;
  // ---- hand over
  ix.slots = pool.keep(d_slots, c.dev_allocs);
  ix.labels = pool.keep(d_dev_labels, c.dev_allocs);
  ix.log2_cap = log2_cap;
  ix.hslots = pool.keep(d_hslots, c.dev_allocs);
  ix.hlist = pool.keep(d_hlist, c.dev_allocs);
  ix.h_log2_cap = hl;
  ix.refp = pool.keep(d_refp, c.dev_allocs);
  ix.pos_flags = pool.keep(d_flags, c.dev_allocs);
  ix.tail_info = pool.keep(d_tail, c.dev_allocs);
  ix.filt[0] = pool.keep(d_f0, c.dev_allocs);
  ix.filt[1] = pool.keep(d_f1, c.dev_allocs);
  ix.hint_first = hint_first;
  ix.n_hint = hint_n;
  ix.filt_log2 = hints ? fl : 0;
  ix.win = n_win ? pool.keep(d_win, c.dev_allocs) : nullptr;
  ix.site_win = n_win ? pool.keep(d_site_win, c.dev_allocs) : nullptr;
  ix.win_base = win_base;
  ix.n_win = n_win;
  c.dev_index = ix;
  c.lookup_tables = {{d_slots, (static_cast<uint64_t>(BUCKET_SLOTS) << log2_cap) * sizeof(IndexSlot)},
                     {d_hslots, (static_cast<uint64_t>(BUCKET_SLOTS) << hl) * sizeof(IndexSlot)},
                     {d_hlist, 2ull * n_keys * sizeof(HalfEntry)},
                     {d_dev_labels, static_cast<uint64_t>(E) * sizeof(DevLabel)}};
  c.n_keys = n_keys;
  c.n_labels = E;
  c.d_keys = pool.keep(d_keys, c.dev_allocs);
  c.d_key_off = pool.keep(d_key_off, c.dev_allocs);
  c.d_labels_sorted = pool.keep(d_labels_sorted, c.dev_allocs);
  c.express4_wide = express4_prefers_wide(c.graph, n_keys, several);
  return GTX_OK;
}

// the device-built index in reference order, for the inspection entry points (gtx_index_get / gtx_index_dump)
int download_index(gtx_ctx & c)
{
  if (c.device < 0)
    return GTX_OK;
  // (several host threads may inspect one context: the first one in fills the vectors, the others wait for it)
  std::lock_guard<std::mutex> lock(c.index_mutex);
  if (c.index_downloaded.load(std::memory_order_acquire))
    return GTX_OK;
  if (!ok_hip(hipSetDevice(c.device), "hipSetDevice"))
    return GTX_ERR_HIP;
  c.index.keys.resize(c.n_keys);
  c.index.key_off.resize(static_cast<size_t>(c.n_keys) + 1);
  c.index.labels.resize(c.n_labels);
  if ((c.n_keys && !ok_hip(hipMemcpy(c.index.keys.data(), c.d_keys, static_cast<size_t>(c.n_keys) * 8, hipMemcpyDeviceToHost), "keys")) ||
      !ok_hip(hipMemcpy(c.index.key_off.data(), c.d_key_off, (static_cast<size_t>(c.n_keys) + 1) * 4, hipMemcpyDeviceToHost), "key offsets") ||
      (c.n_labels && !ok_hip(hipMemcpy(c.index.labels.data(), c.d_labels_sorted, static_cast<size_t>(c.n_labels) * sizeof(gtx_label), hipMemcpyDeviceToHost), "labels")))
    return GTX_ERR_HIP;
  if (c.n_keys == 0)
    c.index.key_off.assign(1, 0);
  c.index_downloaded.store(true, std::memory_order_release);
  return GTX_OK;
}

// one table of the position-hinted pass as the device holds it (inspection: gtx_ctx_hint_table)
int download_hint_table(gtx_ctx const & c, int which, void * out, uint64_t cap_bytes, uint64_t * bytes)
{
  IndexView const & ix = c.dev_index;
  uint64_t const filt_words = ix.n_hint ? (1ull << ix.filt_log2) : 1;
  void const * src = nullptr;
  uint64_t n = 0;
  switch (which)
  {
  case 0: src = ix.pos_flags; n = (ix.n_hint ? hint_total_positions(ix.n_hint, ix.n_win) : 1) * sizeof(uint2_t); break;
  case 1: src = ix.refp; n = (ix.n_hint ? 4 * (hint_total_positions(ix.n_hint, ix.n_win) / 32 + 8) : 32) * sizeof(uint32_t); break;
  case 2: src = ix.tail_info; n = (ix.n_hint ? hint_total_positions(ix.n_hint, ix.n_win) : 1) * sizeof(uint2_t); break;
  case 3: src = ix.filt[0]; n = filt_words * sizeof(uint32_t); break;
  case 4: src = ix.filt[1]; n = filt_words * sizeof(uint32_t); break;
  case 5: src = ix.win; n = static_cast<uint64_t>(ix.n_win) * sizeof(HintWindow); break;
  case 6: src = ix.site_win; n = ix.n_win ? static_cast<uint64_t>(c.graph.ref_order.size()) * sizeof(uint32_t) : 0; break;
  default: return GTX_ERR_ARG;
  }
  *bytes = n;
  if (!out)
    return GTX_OK;
  if (cap_bytes < n)
    return GTX_ERR_CAPACITY;
  if (n != 0 && (!ok_hip(hipSetDevice(c.device), "hipSetDevice") || !ok_hip(hipMemcpy(out, src, n, hipMemcpyDeviceToHost), "hint table")))
    return GTX_ERR_HIP;
  return GTX_OK;
}
} // namespace gtx
