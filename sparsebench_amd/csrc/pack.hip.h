// pack.hip.h -- device-private, LOSSLESS compression of the Sell-C-sigma (C = 64) stream.
//
// The matrix stream is 77 % of the bytes a CG iteration moves, and the reference-layout kernel
// is HBM-bound, so bytes are time.  Levels, each detected automatically at upload and each
// falling back to the previous one when the matrix does not qualify (SB_PACK caps the level):
//
//   1  columns  per chunk, 16-bit offsets from the chunk's smallest column when the chunk's
//               columns span < 65535 (banded / stencil / well-ordered matrices); chunks that
//               do not qualify (e.g. rows touching halo columns) keep 32-bit indices;
//   2  values   if the matrix holds <= 256 distinct fp64 bit patterns (stencils, graph
//               Laplacians, ...): one byte per element indexing a dictionary held in LDS;
//   3  windows  per tile (4 chunks = one workgroup) the x entries its rows touch are staged in
//               LDS; the 16-bit column becomes a slot in that window (spmv_scs64_lds);
//   4  patterns one byte per element names a (value, slot delta) pair of the tile's class;
//   5  rows     a chunk = one shared dominant row pattern + its few odd lanes (spmv_scs64_pat).
//
// Levels 1-2: elements are regrouped so that a lane fetches four consecutive columns of its row
// with one load: 8 B (or 16 B wide) of indices + 4 B of codes per lane per group,
// i.e. 512 B / 1 KiB / 256 B per wave-instruction.  The host-visible arrays keep the
// reference layout (src/SCSMatrix.h); this is a private mirror built once at upload.
// Decoded (column, value) pairs are bit-identical to the originals and are consumed
// in the same left-to-right order, so results do not change by a single bit.
//
// Reference semantics of padding (column 0, value 0.0, src/matrix-SCS.c:146-155) are
// kept: a padded element is encoded as the marker 0xFFFF and decodes to `padCol`
// (column 0, renumbered like any other column when sigma > 1).  (The CRS format's mirror
// uses the SKIPPAD instantiation of the level 4-5 kernel instead: CRS has no padding terms.)
#pragma once
#include "kernels.hip.h"

namespace sbk {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct PackMeta {
  uint32_t grp;    // groups of 4 columns before this chunk (code stream position)
  uint32_t idxOff; // index stream position in 512-B units (a wide group takes two)
  uint32_t base;   // smallest column of the chunk (narrow chunks)
  uint32_t info;   // bit 31: wide (32-bit indices); bits 0..30: chunk width (columns)
};

constexpr uint32_t PACK_PAD = 0xFFFFu;

// ---- analysis / packing (run once at upload) --------------------------------------
__global__ __launch_bounds__(256) void pack_minmax_k(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, uint32_t nChunks, uint32_t padCol, uint32_t* __restrict__ cmin,
    uint32_t* __restrict__ cmax)
{
  const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const uint32_t cp = chunkPtr[chunk], len = chunkLens[chunk];
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  for (uint32_t j = 0; j < len; j++) {
    const uint32_t c = colInd[(size_t)cp + (size_t)j * 64 + lane];
    const double v   = val[(size_t)cp + (size_t)j * 64 + lane];
    const bool pad   = c == padCol && __double_as_longlong(v) == 0;
    if (!pad) {
      lo = min(lo, c);
      hi = max(hi, c);
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
  }
  if (lane == 0) cmin[chunk] = lo, cmax[chunk] = hi;
}

__device__ __forceinline__ uint32_t dict_code(const unsigned long long* __restrict__ dict, int n,
    unsigned long long bits)
{ // dict is sorted by bit pattern; the value is known to be present
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (dict[mid] < bits) lo = mid + 1;
    else hi = mid;
  }
  return (uint32_t)lo;
}

__global__ __launch_bounds__(256) void pack_write_k(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const PackMeta* __restrict__ meta,
    const unsigned long long* __restrict__ dictBits, int nDict, uint32_t nChunks, uint32_t padCol,
    uint32_t* __restrict__ idxOut, uint32_t* __restrict__ codesOut)
{
  const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const uint32_t cp = chunkPtr[chunk], len = chunkLens[chunk];
  const PackMeta m  = meta[chunk];
  const bool wide   = m.info >> 31;
  const uint32_t ng = (len + 3u) >> 2;
  for (uint32_t g = 0; g < ng; g++) {
    uint32_t c[4], code = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t j = g * 4u + k;
      uint32_t col     = padCol;
      double v         = 0.0;
      if (j < len) {
        col = colInd[(size_t)cp + (size_t)j * 64 + lane];
        v   = val[(size_t)cp + (size_t)j * 64 + lane];
      }
      const bool pad = col == padCol && __double_as_longlong(v) == 0;
      c[k]           = wide ? col : (pad ? PACK_PAD : col - m.base);
      if (nDict > 0) code |= dict_code(dictBits, nDict, (unsigned long long)__double_as_longlong(v)) << (8u * k);
    }
    if (wide) {
      uint32_t* o = idxOut + (size_t)m.idxOff * 128 + (size_t)g * 256 + lane * 4; // 16 B per lane
      o[0] = c[0], o[1] = c[1], o[2] = c[2], o[3] = c[3];
    } else {
      uint32_t* o = idxOut + ((size_t)(m.idxOff + g) * 64 + lane) * 2; // 8 B per lane
      o[0] = c[0] | (c[1] << 16), o[1] = c[2] | (c[3] << 16);
    }
    if (nDict > 0) codesOut[(size_t)(m.grp + g) * 64 + lane] = code;
  }
}

// ---- SpMV on the packed stream --------------------------------------------------------
// One wavefront per chunk, lane = row, exactly as spmv_scs64; per group of four columns
// a lane issues one index load, one code load (DICT) and four x gathers.  Two groups
// (eight columns) are kept in flight.  (Deeper batching / prefetching variants were
// measured slower: they cost occupancy, and the limiter is the gather instruction rate
// of the vector cache, not the length of the dependency chain.)
#ifdef SB_LAB // levels 1-2 as a kernel of their own: lab builds only (the BUILD of these levels feeds level 6 and stays)
template <bool DICT, bool DOT>
__global__ __launch_bounds__(256) void spmv_scs64_packed(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ idx, const uint32_t* __restrict__ codes,
    const double* __restrict__ dict, const uint32_t* __restrict__ chunkPtr,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, uint32_t nr,
    uint32_t nChunks, uint32_t blocksPerXcd, uint32_t padCol, double* __restrict__ dotPartials,
    const int* __restrict__ stop)
{
  __shared__ double sdict[256];
  const int stopped      = stop ? *stop : 0;
  const uint32_t nBlocks = (nChunks + 3u) >> 2;
  const uint32_t lb      = blocksPerXcd ? xcd_block(blockIdx.x, blocksPerXcd) : blockIdx.x;
  if (lb >= nBlocks || stopped) return; // uniform per workgroup
  if (DICT) {
    sdict[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
  }
  const uint32_t chunk = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
  const uint32_t lane  = threadIdx.x & 63u;
  const bool active    = chunk < nChunks;
  double acc           = 0.0;
  if (active) {
    const PackMeta m     = meta[chunk];
    const uint32_t cpv   = DICT ? 0u : chunkPtr[chunk];
    const uint32_t len   = m.info & 0x7FFFFFFFu;
    const bool wide      = m.info >> 31;
    const uint32_t ng    = (len + 3u) >> 2;
    const uint32_t* cstream = codes + (size_t)m.grp * 64 + lane;
    const double* vraw      = val + cpv + lane;
    if (!wide) {
      const u32x2* istream = reinterpret_cast<const u32x2*>(idx) + (size_t)m.idxOff * 64 + lane;
      for (uint32_t g = 0; g < ng; g += 2) {
        const bool two = g + 1 < ng; // wave-uniform
        const u32x2 i0 = stream_load(istream + (size_t)g * 64);
        u32x2 i1       = u32x2{ PACK_PAD | (PACK_PAD << 16), PACK_PAD | (PACK_PAD << 16) };
        uint32_t cw0 = 0, cw1 = 0;
        if (two) i1 = stream_load(istream + (size_t)(g + 1) * 64);
        if (DICT) {
          cw0 = stream_load(cstream + (size_t)g * 64);
          if (two) cw1 = stream_load(cstream + (size_t)(g + 1) * 64);
        }
        const uint32_t d[8] = { i0.x & 0xFFFFu, i0.x >> 16, i0.y & 0xFFFFu, i0.y >> 16,
                                i1.x & 0xFFFFu, i1.x >> 16, i1.y & 0xFFFFu, i1.y >> 16 };
        double xv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) xv[k] = x[d[k] == PACK_PAD ? padCol : m.base + d[k]];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t j = g * 4u + (uint32_t)k;
          if (j < len) { // wave-uniform: columns beyond the chunk's width are not accumulated
            const uint32_t cw = k < 4 ? cw0 : cw1;
            const double vv   = DICT ? sdict[(cw >> (8u * (k & 3))) & 255u] : stream_load(vraw + (size_t)j * 64);
            acc               = acc + vv * xv[k];
          }
        }
      }
    } else {
      const u32x4* istream = reinterpret_cast<const u32x4*>(idx + (size_t)m.idxOff * 128) + lane;
      for (uint32_t g = 0; g < ng; g++) {
        const u32x4 i0      = stream_load(istream + (size_t)g * 64);
        const uint32_t cw0  = DICT ? stream_load(cstream + (size_t)g * 64) : 0u;
        const uint32_t d[4] = { i0.x, i0.y, i0.z, i0.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t j = g * 4u + (uint32_t)k;
          if (j < len) {
            const double vv = DICT ? sdict[(cw0 >> (8u * k)) & 255u] : stream_load(vraw + (size_t)j * 64);
            acc             = acc + vv * x[d[k]];
          }
        }
      }
    }
  }
  if (active) spmv_epilogue<DOT>(chunk, lane, acc, x, y, nr, dotPartials);
}
#endif // SB_LAB

// ---- level 3: the x window of a tile staged in LDS -------------------------------------
// With 3 bytes per element the kernel is no longer HBM-bound; PMC shows the vector
// memory pipe busy with the gathers (27 per row, ~16 L1 requests per wave-instruction,
// 90 % L1 hits) and waves waiting on them.  A tile = the 4 chunks (256 rows) of one
// workgroup.  At pack time the host lists, per tile, the few contiguous column ranges
// ("segments") its rows touch -- for a 27-point stencil 3 ranges of ~514 entries, for a
// rank-boundary tile additionally a range of halo columns -- and every element's 16-bit
// code becomes a SLOT in that window (slot 0 always holds x[padCol], so padding needs no
// special case).  The workgroup copies the segments of x into LDS with coalesced loads,
// then every x "gather" is a ds_read_b64 (2 LDS cycles per wave-instruction instead of a
// trip through the vector cache).  Values still come from the dictionary or the fp64
// stream; order and bits are unchanged.
struct TileSeg {
  uint32_t col; // first column of the segment (device numbering)
  uint32_t len; // entries
  uint32_t lds; // first slot in the tile's window
  uint32_t pad_;
};

__global__ __launch_bounds__(256) void pack_slots_k(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ tileSegPtr, const TileSeg* __restrict__ segs, uint32_t nChunks,
    uint32_t padCol, uint32_t* __restrict__ slotOut, uint32_t cpt)
{ // a block = 4 consecutive chunks; they belong to tile (4 * blockIdx) / cpt (cpt = chunks per tile: 4 or 8)
  const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t tile  = chunk / cpt;
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const uint32_t s0 = tileSegPtr[tile], s1 = tileSegPtr[tile + 1];
  const uint32_t cp = chunkPtr[chunk], len = chunkLens[chunk];
  const PackMeta m  = meta[chunk];
  const uint32_t ng = (len + 3u) >> 2;
  for (uint32_t g = 0; g < ng; g++) {
    uint32_t sl[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t j = g * 4u + k;
      uint32_t slot    = 0; // x[padCol]
      if (j < len) {
        const uint32_t col = colInd[(size_t)cp + (size_t)j * 64 + lane];
        const double v     = val[(size_t)cp + (size_t)j * 64 + lane];
        if (!(col == padCol && __double_as_longlong(v) == 0)) {
          uint32_t lo = s0, hi = s1 - 1; // last segment whose first column <= col
          while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (segs[mid].col <= col) lo = mid;
            else hi = mid - 1;
          }
          slot = segs[lo].lds + (col - segs[lo].col);
        }
      }
      sl[k] = slot;
    }
    uint32_t* o = slotOut + ((size_t)(m.grp + g) * 64 + lane) * 2;
    o[0] = sl[0] | (sl[1] << 16), o[1] = sl[2] | (sl[3] << 16);
  }
}

#ifdef SB_LAB // level 3 as a kernel of its own: lab builds only
template <bool DICT, bool DOT>
__global__ __launch_bounds__(256) void spmv_scs64_lds(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ slots, const uint32_t* __restrict__ codes,
    const double* __restrict__ dict, const uint32_t* __restrict__ chunkPtr,
    const double* __restrict__ val, const uint32_t* __restrict__ tileSegPtr,
    const TileSeg* __restrict__ segs, const double* __restrict__ x, double* __restrict__ y, uint32_t nr,
    uint32_t nChunks, uint32_t blocksPerXcd, uint32_t padCol, double* __restrict__ dotPartials,
    const int* __restrict__ stop)
{
  extern __shared__ __attribute__((aligned(16))) double lds[]; // [256 dict][window]
  double* sdict = lds;
  double* sx    = lds + 256;
  constexpr int PF = 8; // groups prefetched before the window is staged (a 32-column chunk)
  const int stopped     = stop ? *stop : 0;
  const uint32_t tile   = blocksPerXcd ? xcd_block(blockIdx.x, blocksPerXcd) : blockIdx.x;
  const uint32_t nTiles = (nChunks + 3u) >> 2;
  if (tile >= nTiles || stopped) return; // uniform per workgroup
  const uint32_t chunk = __builtin_amdgcn_readfirstlane(tile * 4u + (threadIdx.x >> 6));
  const uint32_t lane  = threadIdx.x & 63u;
  const bool active    = chunk < nChunks; // wave-uniform; inactive waves still help staging
  PackMeta m           = { 0u, 0u, 0u, 0u };
  if (active) m = meta[chunk];
  const uint32_t cpv = (DICT || !active) ? 0u : chunkPtr[chunk];
  const uint32_t len = m.info & 0x7FFFFFFFu;
  const uint32_t ng  = (len + 3u) >> 2;
  const u32x2* sstream    = reinterpret_cast<const u32x2*>(slots) + (size_t)m.grp * 64 + lane;
  const uint32_t* cstream = codes + (size_t)m.grp * 64 + lane;
  const double* vraw      = val + cpv + lane;
  // 1. the chunk's slot/code stream goes in flight first (HBM latency) ...
  u32x2 iv[PF];
  uint32_t cw[PF];
#pragma unroll
  for (int gi = 0; gi < PF; gi++) {
    iv[gi] = u32x2{ 0u, 0u }, cw[gi] = 0u;
    if ((uint32_t)gi < ng) {
      iv[gi] = stream_load(sstream + (size_t)gi * 64);
      if (DICT) cw[gi] = stream_load(cstream + (size_t)gi * 64);
    }
  }
  // 2. ... while the workgroup stages its x window (mostly L2 hits)
  if (DICT) sdict[threadIdx.x] = dict[threadIdx.x];
  if (threadIdx.x == 0) sx[0] = x[padCol]; // slot 0: what padding multiplies (src/matrix-SCS.c:151-155)
  const uint32_t s0 = tileSegPtr[tile], s1 = tileSegPtr[tile + 1];
  for (uint32_t s = s0; s < s1; s++) {
    const TileSeg sg = segs[s];
    for (uint32_t i = threadIdx.x; i < sg.len; i += 256u) sx[sg.lds + i] = x[sg.col + i];
  }
  __syncthreads();
  if (!active) return;
  // 3. accumulate left to right: x from LDS, values from the dictionary or the fp64 stream
  double acc = 0.0;
#pragma unroll
  for (int gi = 0; gi < PF; gi++) {
    if ((uint32_t)gi < ng) {
      const uint32_t d[4] = { iv[gi].x & 0xFFFFu, iv[gi].x >> 16, iv[gi].y & 0xFFFFu, iv[gi].y >> 16 };
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t j = (uint32_t)gi * 4u + k;
        if (j < len) {
          const double vv = DICT ? sdict[(cw[gi] >> (8u * k)) & 255u] : stream_load(vraw + (size_t)j * 64);
          acc             = acc + vv * sx[d[k]];
        }
      }
    }
  }
  for (uint32_t g = PF; g < ng; g++) { // chunks wider than 32 columns
    const u32x2 i0      = stream_load(sstream + (size_t)g * 64);
    const uint32_t cw0  = DICT ? stream_load(cstream + (size_t)g * 64) : 0u;
    const uint32_t d[4] = { i0.x & 0xFFFFu, i0.x >> 16, i0.y & 0xFFFFu, i0.y >> 16 };
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t j = g * 4u + k;
      if (j < len) {
        const double vv = DICT ? sdict[(cw0 >> (8u * k)) & 255u] : stream_load(vraw + (size_t)j * 64);
        acc             = acc + vv * sx[d[k]];
      }
    }
  }
  spmv_epilogue<DOT>(chunk, lane, acc, x, y, nr, dotPartials);
}
#endif // SB_LAB

// ---- level 4: pattern dictionary ---------------------------------------------------------
// In the LDS-window kernel an element costs 3 B (value code + 16-bit slot).  In a matrix
// with repeating row patterns (any stencil; any mesh numbered line by line) the PAIR
// (value, slot - slot of the row's first element) takes few distinct values per tile, so
// one byte can name the pair.  Per tile the device collects its distinct pairs; the host
// merges tiles into classes of <= 255 pairs (a 27-point stencil on one GPU: one or a
// handful of classes); a workgroup stages its class's table (4 KiB: value, delta, mask)
// next to the x window and per element does
//        e = table[code];   acc = acc + e.v * window[8 * rowBase * e.m + e.off8]   (byte offset)
// rowBase is a 16-bit slot per row; padding keeps its reference meaning through
// m = 0, off8 = 0 (slot 0 = x[padCol]).  1 B per element instead of 3, still lossless,
// same order, same bits.  Any tile with > 255 pairs: the matrix stays at level 3.
struct PatEntry {
  double v;
  uint32_t off8; // byte offset of the element's x in the window, relative to 8 * rowBase * m
  uint32_t m;    // 1; 0 for padding (then off8 = 0: slot 0)
};

constexpr uint32_t PAT_ABS   = 0x01000000u; // key bit: absolute slot 0 (padding)
constexpr uint32_t PAT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t PAT_MAX   = 255u;        // pairs per class

__device__ __forceinline__ uint32_t pat_key(uint32_t slot, uint32_t base, uint32_t vcode)
{
  return slot == 0u ? (PAT_ABS | vcode) : (vcode | (((slot - base + 32768u) & 0xFFFFu) << 8));
}

// slot / value code of element (row `lane`, column j) from the level-3 streams
__device__ __forceinline__ void pat_fetch(const uint32_t* __restrict__ slots,
    const uint32_t* __restrict__ codes, const PackMeta& m, uint32_t lane, uint32_t j, uint32_t& slot,
    uint32_t& vcode)
{
  const size_t p    = ((size_t)(m.grp + (j >> 2)) * 64 + lane);
  const uint32_t sw = slots[p * 2 + ((j >> 1) & 1u)];
  slot              = (j & 1u) ? sw >> 16 : sw & 0xFFFFu;
  vcode             = (codes[p] >> (8u * (j & 3u))) & 255u;
}

__global__ __launch_bounds__(512) void pat_collect_k(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ slots, const uint32_t* __restrict__ codes, uint32_t nChunks,
    uint16_t* __restrict__ rowBase, uint32_t* __restrict__ tileCount, uint32_t* __restrict__ tileKeys)
{ // one workgroup per tile, one wave per chunk (blockDim = 64 * chunks per tile)
  __shared__ uint32_t table[1024];
  __shared__ uint32_t count, outPos;
  const uint32_t tile  = blockIdx.x;
  const uint32_t chunk = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) table[i] = PAT_EMPTY;
  if (threadIdx.x == 0) count = 0u, outPos = 0u;
  __syncthreads();
  if (chunk < nChunks) {
    const PackMeta m   = meta[chunk];
    const uint32_t len = m.info & 0x7FFFFFFFu;
    uint32_t base = 0u, vc;
    if (len) pat_fetch(slots, codes, m, lane, 0u, base, vc);
    rowBase[(size_t)chunk * 64 + lane] = (uint16_t)base;
    for (uint32_t j = 0; j < len; j++) {
      uint32_t slot, vcode;
      pat_fetch(slots, codes, m, lane, j, slot, vcode);
      const uint32_t key = pat_key(slot, base, vcode);
      uint32_t h         = (key * 2654435761u) >> 22; // 10 bits
      // every thread re-reads count before probing, so the table (1024) never fills:
      // at most PAT_MAX + blockDim (<= 512) distinct keys get in
      while (*(volatile uint32_t*)&count <= PAT_MAX) {
        const uint32_t old = atomicCAS(&table[h], PAT_EMPTY, key);
        if (old == PAT_EMPTY) {
          atomicAdd(&count, 1u);
          break;
        }
        if (old == key) break;
        h = (h + 1u) & 1023u;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) tileCount[tile] = count;
  if (count <= PAT_MAX)
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x)
      if (table[i] != PAT_EMPTY) tileKeys[(size_t)tile * 256 + atomicAdd(&outPos, 1u)] = table[i];
}

__global__ __launch_bounds__(512) void pat_encode_k(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ slots, const uint32_t* __restrict__ codes, uint32_t nChunks,
    const uint16_t* __restrict__ rowBase, const uint32_t* __restrict__ tileClass,
    const uint32_t* __restrict__ classKeys, uint32_t* __restrict__ jcodes)
{
  __shared__ uint32_t keys[256]; // ascending, padded with PAT_EMPTY
  const uint32_t tile  = blockIdx.x; // one workgroup per tile, one wave per chunk
  const uint32_t chunk = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  if (threadIdx.x < 256u) keys[threadIdx.x] = classKeys[(size_t)tileClass[tile] * 256 + threadIdx.x];
  __syncthreads();
  if (chunk >= nChunks) return;
  const PackMeta m    = meta[chunk];
  const uint32_t len  = m.info & 0x7FFFFFFFu;
  const uint32_t base = rowBase[(size_t)chunk * 64 + lane];
  const uint32_t ng   = (len + 3u) >> 2;
  for (uint32_t g = 0; g < ng; g++) {
    uint32_t word = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t j = g * 4u + k;
      if (j < len) {
        uint32_t slot, vcode;
        pat_fetch(slots, codes, m, lane, j, slot, vcode);
        const uint32_t key = pat_key(slot, base, vcode);
        uint32_t lo = 0u, hi = 255u; // the key is present by construction
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (keys[mid] < key) lo = mid + 1u;
          else hi = mid;
        }
        word |= lo << (8u * k);
      }
    }
    jcodes[(size_t)(m.grp + g) * 64 + lane] = word;
  }
}

// ---- level 5: row patterns -------------------------------------------------------------------
// With one byte per element the kernel moves ~98 MB at 128^3 and is bound by its LDS phase
// (16 B table entry + 8 B x per element and lane) and by the code stream.  But in a chunk of
// 64 rows almost all rows carry the SAME code sequence (same values, same window-slot
// deltas): a 27-point stencil chunk has 61-63 such rows and 1-3 odd ones (grid boundary,
// rows the sigma sort moved).  So a chunk is stored as
//     the offset of its dominant ROW PATTERN  -- len entries (value, delta, mask), shared by
//                                                all chunks with that pattern (a handful)
//     a 64-bit mask of exception lanes + the code sequences of those lanes only.
// Dominant lanes take their entries from scalar registers (s_load through the scalar cache:
// no LDS, no per-lane stream); exception lanes use their codes and the tile's table in LDS
// as before and override per lane.  Chunks with many exceptions, or once the pattern table
// is full, stay in the per-lane form (L chunks).  Lossless, same order, same bits.
constexpr uint32_t PAT_INLINE_SEGS = 6;
constexpr uint32_t PAT_UNIFORM     = 0x80000000u; // TileHdr.len[] flag: dominant-pattern chunk
constexpr uint32_t PAT_NOPAD       = 0x40000000u; // ... whose dominant pattern holds no padding (masked form: no masked lane)
constexpr uint32_t PAT_HASPAD      = 0x20000000u; // masked form: some row of the chunk is shorter than the chunk (reference padding)
constexpr uint32_t PAT_LEN_MASK    = 0x1FFFFFFFu;
constexpr uint32_t PAT_EXC_MAX     = 8; // more exception lanes than this: L chunk (an exception
                                        // row costs 16 B per column, a per-lane row 1 B)

// Everything a workgroup needs to know about its tile, in one 192-byte record: each round of
// tiles a CU runs pays every DEPENDENT memory round trip once, so header -> {codes, row
// bases, tables, x window} is the whole chain: two round trips, everything else in parallel.
struct TileHdr {
  uint32_t cls;       // pattern class
  uint32_t nseg;      // segments of the window
  uint32_t segPtr;    // first segment in the global list (tiles with > PAT_INLINE_SEGS)
  uint32_t win;       // window entries incl. slot 0
  uint32_t off[4];    // L chunk: code-stream position (words); U chunk: first exception entry
  uint32_t len[4];    // chunk widths (0 for chunks past the end) | PAT_UNIFORM | PAT_NOPAD
  uint32_t seg[PAT_INLINE_SEGS][3]; // first column, first slot (0xFFFFFFFF: unused), entries
  uint32_t winInline; // window entries covered by the inline segments
  uint32_t flags;     // PAT_SIMPLE_WINDOW
  uint32_t rowPat[4]; // U chunks: first entry of the dominant row pattern
  uint32_t exc[4][2]; // U chunks: exception lanes (lo, hi)
  uint32_t excStart;  // the tile's exception entries: first, count (contiguous over its U chunks)
  uint32_t excCount;
  uint32_t tile;      // which tile this is: headers are stored interior tiles first (see below)
  uint32_t pad_;
};
static_assert(sizeof(TileHdr) == 192, "TileHdr is 48 words");
constexpr int PAT_STOP_LANE          = 48; // the lane that fetches the stop flag next to the header
// <= 6 segments, listed longest first: three of <= 768 entries (3 loads per thread) and three
// of <= 256 (1 load): staged segment by segment, no per-entry search
constexpr uint32_t PAT_SIMPLE_WINDOW = 1u;
constexpr uint32_t PAT_TOUCHES_HALO  = 2u; // the window holds columns >= nr (entries of other ranks)
// level 6, row-permuted matrices: the window is laid out in ORIGINAL column order and staged through a 16-bit
// map (slot -> device column - base of the slot's block of 256; the 18 segment words hold the block bases)
constexpr uint32_t PAT_MAPPED_WINDOW = 4u;

// HALO instantiation of spmv_scs64_pat (several ranks, peer-mapped halo exchange, kernels.hip.h):
// the neighbours' halo_push_k kernels store straight into this rank's staging area; instead of a
// separate pull launch, the few tiles that touch halo columns wait for the sources' sequence flags
// themselves and read those columns from the staging area.  They are stored last (interior tiles
// first), so by the time they run the data is normally there: the exchange hides behind the
// interior tiles inside ONE kernel, without a second stream.
struct HaloWait {
  const unsigned long long* flags; // own flags, [2][P2P_MAX]
  const double* ext;               // own staging area of this exchange's parity: column c >= nr is ext[c - nr]
  const int* src;                  // source ranks
  int nsrc;
  unsigned long long seq;
  int* err;
  int* stopw;             // CgScalars::stop, raised together with err: no iterating on a stale halo
  long long timeoutTicks; // bound of the wait (HaloPush::timeoutTicks)
  // the rank's own push rides in the same launch: its first nPush workgroups send x[elementsToSend] to the neighbours
  // (kernels.hip.h: halo_push_block) instead of multiplying a tile -- they are dispatched first, the tiles that wait
  // for the neighbours' pushes last.  nPush = 0: a separate halo_push_k has been launched.
  const HaloPush* push; // device copy
  uint32_t nPush;       // a multiple of 8 (keeps the workgroup -> XCD assignment of the tiles)
};
constexpr uint32_t PAT_EXC_LDS_MAX   = 1024; // exception entries per 4 chunks of a tile (16 KiB of LDS) at most

// dominant code sequence of every chunk (majority of the 64 lanes) and the lanes that differ
__global__ __launch_bounds__(256) void pat_dominant_k(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ jcodes, uint32_t nChunks, uint32_t* __restrict__ domCodes,
    uint32_t* __restrict__ excMask)
{
  const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const PackMeta m  = meta[chunk];
  const uint32_t ng = ((m.info & 0x7FFFFFFFu) + 3u) >> 2;
  const uint32_t* c = jcodes + (size_t)m.grp * 64 + lane;
  uint32_t h        = 2166136261u;
  for (uint32_t g = 0; g < ng; g++) h = (h ^ c[(size_t)g * 64]) * 16777619u;
  uint32_t cnt = 0;
  for (int k = 0; k < 64; k++) cnt += h == (uint32_t)__builtin_amdgcn_readlane((int)h, k);
  uint32_t best = (cnt << 6) | (63u - lane); // most frequent hash, lowest lane on ties
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) best = max(best, (uint32_t)__shfl_xor((int)best, off, 64));
  const int dom = 63 - (int)(best & 63u);
  bool same     = true; // exact comparison with the dominant lane (a hash can collide)
  for (uint32_t g = 0; g < ng; g++) {
    const uint32_t w  = c[(size_t)g * 64];
    const uint32_t wd = (uint32_t)__shfl((int)w, dom, 64);
    same              = same && w == wd;
    if (lane == 0) domCodes[m.grp + g] = wd;
  }
  const unsigned long long exc = __ballot(!same);
  if (lane == 0) excMask[2 * (size_t)chunk] = (uint32_t)exc, excMask[2 * (size_t)chunk + 1] = (uint32_t)(exc >> 32);
}

// per-lane code words -> final form: L chunks keep the code words of all 64 lanes
// (group-major); U chunks keep only their exception lanes, EXPANDED to (value, byte offset of
// x in the window) entries, `len` per lane, lane after lane
__global__ __launch_bounds__(256) void pat_compact_k(const PackMeta* __restrict__ meta,
    const uint32_t* __restrict__ jcodes, uint32_t nChunks, const uint32_t* __restrict__ chunkOff,
    const uint32_t* __restrict__ chunkFlags, const uint32_t* __restrict__ excMask,
    const uint16_t* __restrict__ rowBase, const uint32_t* __restrict__ tileClass,
    const PatEntry* __restrict__ classDict, uint32_t* __restrict__ stream, PatEntry* __restrict__ excRows,
    uint32_t cpt)
{
  const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const PackMeta m   = meta[chunk];
  const uint32_t len = m.info & 0x7FFFFFFFu, ng = (len + 3u) >> 2;
  const uint32_t* c  = jcodes + (size_t)m.grp * 64 + lane;
  const uint32_t off = chunkOff[chunk];
  if (chunkFlags[chunk] & PAT_UNIFORM) {
    const uint32_t lo = excMask[2 * (size_t)chunk], hi = excMask[2 * (size_t)chunk + 1];
    const bool isExc  = ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u) != 0u;
    const uint32_t ix = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    if (isExc && excRows) { // (masked form: U chunks store nothing per lane)
      const PatEntry* cd   = classDict + (size_t)tileClass[chunk / cpt] * 256;
      const uint32_t base8 = (uint32_t)rowBase[(size_t)chunk * 64 + lane] << 3;
      for (uint32_t j = 0; j < len; j++) {
        const PatEntry e = cd[(c[(size_t)(j >> 2) * 64] >> (8u * (j & 3u))) & 255u];
        excRows[(size_t)off + (size_t)ix * len + j] = PatEntry{ e.v, base8 * e.m + e.off8, e.m };
      }
    }
  } else {
    for (uint32_t g = 0; g < ng; g++) stream[(size_t)off + (size_t)g * 64 + lane] = c[(size_t)g * 64];
  }
}

// ---- level 6: masked row programs --------------------------------------------------------------
// Round-2 measurements of the level-5 kernel: ~1270 VALU clocks per chunk against ~330 for the
// arithmetic, most of the difference spent on the odd lanes -- nearly every chunk has one or two (rows next to
// a grid boundary), so nearly every chunk runs the divergent "override from LDS" branch for every column, and
// their expanded entries cost registers (16 VGPRs of prefetch) and LDS (occupancy).  But an odd row of a
// structured matrix is almost always a SUB-SEQUENCE of a longer row of the same tile, shifted: the stencil row
// at x = 0 is the interior row without its x-1 entries.  So a chunk is stored as a ROW PROGRAM
//     entry j = (value, window offset relative to the lane's base slot, 64-bit mask of the lanes that HAVE it)
// shared by all chunks with the same program, and a per-row 16-bit base slot chosen so that the row's own
// entries line up with the program's.  Every lane executes every entry; the add of entry j runs under
// EXEC = mask j (two scalar moves), so a lane that lacks the entry keeps its sum: no divergence, no per-lane
// data, no exception area in LDS.  The order of a row's additions is unchanged (a sub-sequence of the
// program, left to right), so results are the reference's bit for bit.  Reference padding (Sell-C-sigma rows
// shorter than their chunk add 0.0 * x[padCol] per missing column, src/matrix-SCS.c:151-155 / :208-227) is
// one masked add of 0.0 * x[padCol] behind the program: adding it once or several times gives the same bits
// (s + 0.0 = s for every finite or infinite s that an accumulation starting from +0.0 can hold; NaN stays NaN).
// Chunks whose rows do not fit any program of their tile keep the per-lane code words (L chunks).
// A program is stored in blocks of 8 entries, structure-of-arrays, so that a batch of 8 entries is three scalar
// loads (values, masks, offsets); entries behind the program's end have mask 0.
struct ProgBlock {
  double v[8];
  unsigned long long mask[8]; // lanes that have the entry
  uint32_t off8[8];           // byte offset of the entry's x in the window, relative to 8 * the row's base slot
  uint32_t pad_[8];
};
static_assert(sizeof(ProgBlock) == 192, "blocks stay 64-byte aligned");
typedef double f64x8 __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// acc += prod in the lanes of `mask` only.  EXEC is set for the one add and put back to all ones: the caller is in
// wave-uniform control flow with all 64 lanes active (spmv_scs64_pat: full waves of a 256-thread workgroup, every
// branch around the accumulate loop is on scalar values), which the compiler does not know and need not.
__device__ __forceinline__ void masked_add(double& acc, double prod, unsigned long long mask)
{
  asm volatile("s_mov_b64 exec, %2\n\tv_add_f64 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(acc) : "v"(prod), "s"(mask));
}

// the first N (8 or 4) entries of a program block: entry loads (scalar), x reads (LDS), then products and adds
template <int N, bool MASK>
__device__ __forceinline__ void prog_batch(const ProgBlock* __restrict__ blk, uint32_t base8x, const char* ldsBytes,
    double& acc)
{
  double v[N], xs[N];
  unsigned long long mk[N];
  uint32_t o[N];
  if (N == 8) { // uniform addresses: s_load_dwordx16 / x16 / x8
    const f64x8 vv = *reinterpret_cast<const f64x8*>(blk->v);
    const u32x8 oo = *reinterpret_cast<const u32x8*>(blk->off8);
#pragma unroll
    for (int q = 0; q < N; q++) v[q] = vv[q], o[q] = oo[q];
    if (MASK) {
      const u64x8 mm = *reinterpret_cast<const u64x8*>(blk->mask);
#pragma unroll
      for (int q = 0; q < N; q++) mk[q] = mm[q];
    }
  } else {
    const f64x4 vv = *reinterpret_cast<const f64x4*>(blk->v);
    const u32x4 oo = *reinterpret_cast<const u32x4*>(blk->off8);
    const u64x4 mm = *reinterpret_cast<const u64x4*>(blk->mask);
#pragma unroll
    for (int q = 0; q < N; q++) v[q] = vv[q], o[q] = oo[q], mk[q] = mm[q];
  }
#pragma unroll
  for (int q = 0; q < N; q++) xs[q] = *reinterpret_cast<const double*>(ldsBytes + (base8x + o[q]));
#pragma unroll
  for (int q = 0; q < N; q++) {
    const double prod = v[q] * xs[q];
    if (MASK) masked_add(acc, prod, mk[q]);
    else acc = acc + prod;
  }
}

// Measured (rocprofv3 SQ counters, 128^3): with the stream down to ~40 MB the kernel is bound
// by VALU ISSUE -- 503 vector instructions per wave against ~85 useful ones (address, multiply,
// add per element).  So everything that is not the accumulate loop is kept out of the vector
// ALU: the header is decoded with scalar ops, the window is copied segment by segment (no
// per-entry search), U chunks read no code stream, and their odd lanes get ready-made
// (value, offset) entries from LDS instead of decoding codes.
// SKIPPAD: padded elements are NOT added (the reference's Sell-C-sigma loop adds 0.0 * x[0]
// for them, src/matrix-SCS.c:151-155 / :208-227; its CRS loop has no such elements,
// src/matrix-CRS.c:46-65) -- the instantiation behind the CRS format's private mirror.
// CPT: chunks per tile, 4 or 8.  With 8 a wave multiplies two chunks (w and w + 4 of the tile) behind ONE header
// fetch, one window staging and one barrier: those phases take about as long for 8 chunks as for 4 (measured,
// DESIGN 4.2), the window holds 25 % fewer entries per row, and a CU gets through its tiles in fewer rounds.  The
// 8-chunk header is two 48-word halves interleaved word by word (X: the tile-level fields and chunks 0-3, Y: the
// per-chunk fields of chunks 4-7 in the same positions), fetched as ONE 8-byte vector load.
// MASKED: the level-6 form (row programs in `progs`, signed row bases, no exception entries; rowPats / excRows unused).
template <int CPT, bool DOT, bool SKIPPAD, bool HALO, bool MASKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(72))) /* see spmv_prog_fusep: the HALO instantiations used 83-95 */ void spmv_scs64_pat(const uint32_t* __restrict__ hdrWords,
    const uint32_t* __restrict__ stream, const uint16_t* __restrict__ rowBase,
    const PatEntry* __restrict__ classDict, const PatEntry* __restrict__ rowPats,
    const PatEntry* __restrict__ excRows, const ProgBlock* __restrict__ progs, const uint16_t* __restrict__ slotMap,
    uint32_t mapStride, const TileSeg* __restrict__ segs,
    const double* __restrict__ x, double* __restrict__ y, uint32_t nr, uint32_t nChunks,
    uint32_t firstHdr, uint32_t nHdrs, uint32_t blocksPerXcd, uint32_t padCol, uint32_t dictEntries,
    uint32_t excLds, double* __restrict__ dotPartials, const int* __restrict__ stop, HaloWait hw)
{
  extern __shared__ __attribute__((aligned(16))) double lds[]; // [dict][exception entries + 8][window]
  PatEntry* sd = reinterpret_cast<PatEntry*>(lds);
  PatEntry* se = sd + dictEntries;
  double* sx   = reinterpret_cast<double*>(se + excLds + 8u);
  constexpr int CW   = CPT / 4;          // chunks per wave
  constexpr int PF   = 8;                // code groups prefetched (32 columns); wider chunks stream the rest
  constexpr int LONG = CPT == 8 ? 4 : 3; // loads per thread for each of the three long segments of a simple window
  constexpr int WB   = 3 * LONG + 3;     // window entries per thread in the first pass
  constexpr int EXL  = MASKED ? 0 : CPT / 2; // unconditional exception loads per thread (256 entries each)
  // A launch covers headers [firstHdr, firstHdr + nHdrs).  Headers are stored with the tiles
  // that touch no halo column first, so that on several ranks the interior part of the
  // product can run while the halo is still in flight (one launch for each part).
  if (HALO && blockIdx.x < hw.nPush) { // (uniform per workgroup) this workgroup carries the rank's halo push
    if (*stop) {
      if (blockIdx.x == 0 && threadIdx.x == 0 && halo_rank_failed(*hw.push)) halo_poison_flags(*hw.push); // this rank has failed
      return;
    }
    halo_push_block(*hw.push, x, hw.seq, blockIdx.x, hw.nPush);
    return;
  }
  const uint32_t bid   = HALO ? blockIdx.x - hw.nPush : blockIdx.x;
  const uint32_t tile0 = blocksPerXcd ? xcd_block(bid, blocksPerXcd) : bid;
  const uint32_t hidx  = firstHdr + min(tile0, nHdrs - 1u); // clamped: every load below is unconditional
  // round trip 1: ONE vector load brings the tile header (lanes 0..47) and the stop flag
  // (lane 48); fields are then read out of the lanes (v_readlane -> SGPRs)
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t hvx, hvy = 0u;
  if (CPT == 4) {
    const uint32_t* hp = hdrWords + (size_t)hidx * 48u;
    hvx = *(lane < (uint32_t)PAT_STOP_LANE ? hp + lane : reinterpret_cast<const uint32_t*>(stop));
  } else {
    const u32x2* hp = reinterpret_cast<const u32x2*>(hdrWords + (size_t)hidx * 128u);
    const u32x2 h2  = *(lane < (uint32_t)PAT_STOP_LANE ? hp + lane : reinterpret_cast<const u32x2*>(stop));
    hvx = h2.x, hvy = h2.y;
  }
  // (level 6, mapped windows) the slot map is stored in HEADER order.  Round 3 measured fetching it NEXT TO the header
  // (one dependent hop fewer: header | map -> x instead of header -> map -> x): 0.1-0.3 us of 18.8 (the hop is not what a
  // tile's life is made of), for 3-5 more live registers -- which takes the HALO instantiation from 61 to 66 VGPRs, i.e.
  // from 8 to 7 resident workgroups per CU on every rank of a multi-GPU run.  So the map is fetched behind the header.
  uint32_t dmap[WB];
  auto field = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)hvx, i); };
  // per-chunk field of the wave's c-th chunk (chunk wv + 4 c of the tile): half X for c = 0, half Y for c = 1
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto cfield = [&](int c, int i) -> uint32_t {
    return (uint32_t)__builtin_amdgcn_readlane((int)(c == 0 ? hvx : hvy), i);
  };
  const int stopped    = (int)field(PAT_STOP_LANE);
  const uint32_t tile  = field(46);
  const uint32_t cls = field(0), nseg = field(1), winInline = field(30), flags = field(31);
  const uint32_t excStart = field(44), excCount = field(45);
  const bool simple = (flags & PAT_SIMPLE_WINDOW) != 0u; // uniform per workgroup
  if (HALO && (flags & PAT_TOUCHES_HALO) && !stopped && tile0 < nHdrs) { // wait for the neighbours' blocks
    if ((int)threadIdx.x < hw.nsrc) {
      const unsigned long long* f = hw.flags + (hw.seq & 1ull) * P2P_MAX + hw.src[threadIdx.x];
      const int how               = p2p_wait(f, hw.seq, hw.timeoutTicks);
      if (how) halo_wait_failed(how, hw.err, hw.stopw);
    }
    __syncthreads();
  }
  auto xcol = [&](uint32_t col) -> double { // x by column; halo columns come from the staging area
    if (HALO) return *(col >= nr ? hw.ext + (col - nr) : x + col);
    return x[col];
  };
  // round trip 2: codes (L chunks), row bases, own x entries, tables, x window -- addresses
  // clamped into valid memory so that nothing waits for a branch
  uint32_t chunk[CW], row[CW], off[CW], lenf[CW];
  int32_t base[CW];
  double xrow[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    chunk[c] = tile * CPT + wv + 4u * (uint32_t)c;
    row[c]   = chunk[c] * 64u + lane;
    off[c]   = cfield(c, 4 + (int)wv);
    lenf[c]  = cfield(c, 8 + (int)wv);
    base[c]  = MASKED ? (int32_t)reinterpret_cast<const int16_t*>(rowBase)[chunk[c] < nChunks ? row[c] : 0u]
                      : (int32_t)rowBase[chunk[c] < nChunks ? row[c] : 0u];
    xrow[c]  = DOT ? x[min(row[c], nr - 1u)] : 0.0;
  }
  // (per-lane code words are prefetched for the wave's FIRST chunk only; a second L chunk fetches its own later:
  //  matrices that come here in the 8-chunk form are made of U chunks almost everywhere)
  const bool uni0       = (lenf[0] & PAT_UNIFORM) != 0u; // wave-uniform
  const uint32_t len0   = lenf[0] & PAT_LEN_MASK, ng0 = (len0 + 3u) >> 2;
  const uint32_t gLast0 = ng0 ? ng0 - 1u : 0u;
  uint32_t cw[PF];
#pragma unroll
  for (int gi = 0; gi < PF; gi++) cw[gi] = 0u;
  if (!uni0) {
    const uint32_t* cb0 = stream + (size_t)off[0] + lane;
#pragma unroll
    for (int gi = 0; gi < PF; gi++) cw[gi] = stream_load(cb0 + (size_t)min((uint32_t)gi, gLast0) * 64);
  }
  PatEntry mine = { 0.0, 0u, 0u };
  if (dictEntries) mine = classDict[(size_t)cls * 256 + threadIdx.x];
  PatEntry ex[EXL + 1];
#pragma unroll
  for (int q = 0; q < EXL; q++) ex[q] = excRows[(size_t)excStart + (uint32_t)q * 256u + threadIdx.x]; // (slack behind excRows)
  const double xpad = x[padCol]; // slot 0: what padding multiplies (src/matrix-SCS.c:151-155)
  double t[WB];
  const bool mappedWin = MASKED && (flags & PAT_MAPPED_WINDOW) != 0u; // uniform per workgroup
  if (mappedWin) { // slot by slot through the map
    // (distinct maps stored once behind a header -> map table: sbhip_matrix.inc.h; the index is a uniform load that is back with the header)
    const uint32_t mStr = mapStride & 0xFFFu, mOff = (mapStride >> 12) * 256u;
    const uint32_t mIdx = mOff ? reinterpret_cast<const uint32_t*>(slotMap)[hidx] : hidx;
    const uint16_t* mp  = slotMap + mOff + (size_t)mIdx * mStr + threadIdx.x;
#pragma unroll
    for (int k = 0; k < WB; k++) dmap[k] = mp[min((uint32_t)k * 256u, mStr - 256u)];
#pragma unroll
    for (int k = 0; k < WB; k++) t[k] = xcol(field(12 + min(k, 17)) + dmap[k]);
  } else if (simple) { // segment by segment: entry i of segment s -> slot first_s + i
#pragma unroll
    for (int sI = 0; sI < 3; sI++) {
      const uint32_t sc = field(12 + 3 * sI), sn = field(12 + 3 * sI + 2);
#pragma unroll
      for (int r = 0; r < LONG; r++) t[sI * LONG + r] = xcol(sn ? sc + min((uint32_t)r * 256u + threadIdx.x, sn - 1u) : padCol);
    }
#pragma unroll
    for (int sI = 3; sI < 6; sI++) {
      const uint32_t sc = field(12 + 3 * sI), sn = field(12 + 3 * sI + 2);
      t[3 * LONG + sI - 3] = xcol(sn ? sc + min(threadIdx.x, sn - 1u) : padCol);
    }
  } else { // slot by slot over the inline segments
    uint32_t segCol[PAT_INLINE_SEGS], segFirst[PAT_INLINE_SEGS];
#pragma unroll
    for (int s = 0; s < (int)PAT_INLINE_SEGS; s++) segCol[s] = field(12 + 3 * s), segFirst[s] = field(12 + 3 * s + 1);
#pragma unroll
    for (int k = 0; k < WB; k++) {
      const uint32_t slot = min((uint32_t)k * 256u + threadIdx.x, winInline - 1u);
      uint32_t col        = padCol;
#pragma unroll
      for (uint32_t s = 0; s < PAT_INLINE_SEGS; s++) col = slot >= segFirst[s] ? segCol[s] + (slot - segFirst[s]) : col;
      t[k] = xcol(col);
    }
  }
  // keep every load above in front of the exit test (the compiler would sink them behind it)
#pragma unroll
  for (int k = 0; k < WB; k++) asm volatile("" ::"v"(t[k]));
#pragma unroll
  for (int gi = 0; gi < PF; gi++) asm volatile("" ::"v"(cw[gi]));
#pragma unroll
  for (int c = 0; c < CW; c++) asm volatile("" ::"v"(base[c]), "v"(xrow[c]));
#pragma unroll
  for (int q = 0; q < EXL; q++) asm volatile("" ::"v"(ex[q].v), "v"(ex[q].off8));
  asm volatile("" ::"v"(xpad), "v"(mine.v), "v"(mine.off8), "v"(mine.m));
  if (tile0 >= nHdrs || stopped) return; // uniform per workgroup
  if (mappedWin) {
    const uint32_t win = field(3);
#pragma unroll
    for (int k = 0; k < WB; k++) {
      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;
      if (slot < win) sx[slot] = t[k];
    }
    if (threadIdx.x == 0) sx[0] = xpad;
    if (win > 256u * WB) __builtin_trap(); // (the host builds no such window)
  } else if (simple) {
    if (threadIdx.x == 0) sx[0] = xpad;
#pragma unroll
    for (int sI = 0; sI < 3; sI++) {
      const uint32_t sf = field(12 + 3 * sI + 1), sn = field(12 + 3 * sI + 2);
#pragma unroll
      for (int r = 0; r < LONG; r++) {
        const uint32_t i = (uint32_t)r * 256u + threadIdx.x;
        if (i < sn) sx[sf + i] = t[sI * LONG + r];
      }
    }
#pragma unroll
    for (int sI = 3; sI < 6; sI++) {
      const uint32_t sf = field(12 + 3 * sI + 1), sn = field(12 + 3 * sI + 2);
      if (threadIdx.x < sn) sx[sf + threadIdx.x] = t[3 * LONG + sI - 3];
    }
  } else {
    uint32_t segCol[PAT_INLINE_SEGS], segFirst[PAT_INLINE_SEGS];
#pragma unroll
    for (int s = 0; s < (int)PAT_INLINE_SEGS; s++) segCol[s] = field(12 + 3 * s), segFirst[s] = field(12 + 3 * s + 1);
#pragma unroll
    for (int k = 0; k < WB; k++) {
      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;
      if (slot < winInline) sx[slot] = t[k];
    }
    for (uint32_t w0 = 256u * WB; w0 < winInline; w0 += 256u) { // bigger windows
      const uint32_t slot = w0 + threadIdx.x;
      uint32_t col        = padCol;
#pragma unroll
      for (uint32_t s = 0; s < PAT_INLINE_SEGS; s++) col = slot >= segFirst[s] ? segCol[s] + (slot - segFirst[s]) : col;
      if (slot < winInline) sx[slot] = xcol(col);
    }
    const uint32_t segPtr = field(2);
    for (uint32_t s = PAT_INLINE_SEGS; s < nseg; s++) { // rare: tiles with many ranges
      const TileSeg sg = segs[segPtr + s];
      for (uint32_t i = threadIdx.x; i < sg.len; i += 256u) sx[sg.lds + i] = xcol(sg.col + i);
    }
  }
  // offsets become LDS byte addresses here (once per staged entry, not once per use)
  const uint32_t sxOff = (dictEntries + excLds + 8u) * (uint32_t)sizeof(PatEntry);
#pragma unroll
  for (int q = 0; q < EXL; q++) {
    const uint32_t i = (uint32_t)q * 256u + threadIdx.x;
    if (i < excCount) se[i] = PatEntry{ ex[q].v, ex[q].off8 + sxOff, ex[q].m };
  }
  if (!MASKED)
    for (uint32_t i = (uint32_t)EXL * 256u + threadIdx.x; i < excCount; i += 256u) {
      const PatEntry e = excRows[(size_t)excStart + i];
      se[i]            = PatEntry{ e.v, e.off8 + sxOff, e.m };
    }
  if (dictEntries) sd[threadIdx.x] = PatEntry{ mine.v, mine.off8 + sxOff, mine.m };
  __syncthreads();
  // An element costs: entry -> byte offset of its x in the window -> x -> multiply -> add.
  // Offsets are pre-scaled (base8 * m + off8, m = 0 for padding), and only a chunk's last,
  // partial group pays for the "column < width" selects.
  auto xread = [&](uint32_t o) -> double { // o: LDS byte address
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds) + o);
  };
  double pl0[CW]; // (DOT, MASKED) the wave's level-0 partials of x . y; +0.0 for chunks past the end
#pragma unroll
  for (int c = 0; c < CW; c++) pl0[c] = 0.0;
#pragma unroll
  for (int c = 0; c < CW; c++) {
    if (chunk[c] >= nChunks) continue; // wave-uniform; inactive waves only helped staging
    const uint32_t len = lenf[c] & PAT_LEN_MASK, ng = (len + 3u) >> 2;
    const bool uni     = (lenf[c] & PAT_UNIFORM) != 0u; // wave-uniform
    double acc           = 0.0;
    const uint32_t base8 = (uint32_t)base[c] << 3;
    const uint32_t base8x = base8 + sxOff;
    if (MASKED && uni) {
      // the chunk's row program out of scalar registers; every lane runs every entry, the adds under the entry's mask
      const ProgBlock* pg  = progs + cfield(c, 32 + (int)wv);
      const bool nomask    = (lenf[c] & PAT_NOPAD) != 0u; // wave-uniform: every lane has every entry
      const char* ldsBytes = reinterpret_cast<const char*>(lds);
      uint32_t j0          = 0;
      // (not unrolled: a second batch in flight costs 24-40 scalar and 24 vector registers, i.e. occupancy)
      if (nomask) {
#pragma unroll 1
        for (; j0 + 8u <= len; j0 += 8u) prog_batch<8, false>(pg + (j0 >> 3), base8x, ldsBytes, acc);
      } else {
#pragma unroll 1
        for (; j0 + 8u <= len; j0 += 8u) prog_batch<8, true>(pg + (j0 >> 3), base8x, ldsBytes, acc);
      }
      if (j0 + 4u < len) prog_batch<8, true>(pg + (j0 >> 3), base8x, ldsBytes, acc); // 5..7 entries left (the rest: mask 0)
      else if (j0 < len) prog_batch<4, true>(pg + (j0 >> 3), base8x, ldsBytes, acc); // 1..4
      if (!SKIPPAD && (lenf[c] & PAT_HASPAD)) { // the reference's padding terms; lanes: the header's exc words
        const unsigned long long padMask =
            (unsigned long long)cfield(c, 36 + 2 * (int)wv) | ((unsigned long long)cfield(c, 37 + 2 * (int)wv) << 32);
        masked_add(acc, 0.0 * xpad, padMask);
      }
    } else if (uni) {
      // dominant lanes: entries of the row pattern from scalar registers (s_load through the
      // scalar cache); exception lanes: their own ready-made entries from LDS.  Two groups
      // (8 columns) at a time, so that 8 entry loads / 8 x reads are in flight together.
      const uint32_t excLo = cfield(c, 36 + 2 * (int)wv), excHi = cfield(c, 37 + 2 * (int)wv);
      const bool isExc      = ((lane < 32u ? excLo >> lane : excHi >> (lane - 32u)) & 1u) != 0u;
      const uint32_t excIdx = __builtin_amdgcn_mbcnt_hi(excHi, __builtin_amdgcn_mbcnt_lo(excLo, 0u));
      const PatEntry* rp = rowPats + cfield(c, 32 + (int)wv);
      const PatEntry* me = se + ((off[c] - excStart) + excIdx * len); // this lane's row, if it is an exception
      const bool nopad   = (lenf[c] & PAT_NOPAD) != 0u; // wave-uniform: no padding in the dominant pattern
      auto upair = [&](uint32_t j0, const bool full, const bool np) {
        double v[8], xs[8];
        uint32_t o[8], keep[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
          const PatEntry e = rp[full ? j0 + q : min(j0 + q, len - 1u)]; // uniform address: s_load
          keep[q]          = np ? 1u : e.m;
          // (pinned to scalar registers: otherwise the compiler merges this load with the odd
          //  lanes' LDS read below into ONE flat load through a selected generic pointer)
          const unsigned long long vb = __builtin_bit_cast(unsigned long long, e.v);
          const unsigned long long vs =
              (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vb) |
              ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(vb >> 32)) << 32);
          v[q] = __builtin_bit_cast(double, vs);
          o[q] = np ? base8x + e.off8 : __umul24(base8, e.m) + (e.off8 + sxOff);
        }
        if (isExc) { // divergent: only the odd lanes
#pragma unroll
          for (uint32_t q = 0; q < 8; q++) {
            const PatEntry e = me[j0 + q]; // entries past the row's end belong to the next row / the slack
            v[q]             = e.v;
            o[q]             = e.off8;
            if (SKIPPAD) keep[q] = e.m;
          }
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) xs[q] = xread(o[q]);
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
          const double prod = v[q] * xs[q];
          const double sum  = acc + prod;
          acc               = ((full || j0 + q < len) && (!SKIPPAD || keep[q] != 0u)) ? sum : acc;
        }
      };
      auto uquad = [&](uint32_t j0) { // a chunk's last 1..4 columns
        double v[4], xs[4];
        uint32_t o[4], keep[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const PatEntry e = rp[min(j0 + q, len - 1u)];
          keep[q]          = e.m;
          const unsigned long long vb = __builtin_bit_cast(unsigned long long, e.v);
          const unsigned long long vs =
              (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vb) |
              ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(vb >> 32)) << 32);
          v[q] = __builtin_bit_cast(double, vs);
          o[q] = __umul24(base8, e.m) + (e.off8 + sxOff);
        }
        if (isExc) {
#pragma unroll
          for (uint32_t q = 0; q < 4; q++) {
            const PatEntry e = me[j0 + q];
            v[q]             = e.v;
            o[q]             = e.off8;
            if (SKIPPAD) keep[q] = e.m;
          }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) xs[q] = xread(o[q]);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const double prod = v[q] * xs[q];
          const double sum  = acc + prod;
          acc               = (j0 + q < len && (!SKIPPAD || keep[q] != 0u)) ? sum : acc;
        }
      };
      uint32_t j0 = 0;
      for (; j0 + 8u <= len; j0 += 8u) {
        if (nopad) upair(j0, true, true);
        else upair(j0, true, false);
      }
      if (j0 + 4u < len) upair(j0, false, false); // 5..7 columns left
      else if (j0 < len) uquad(j0);              // 1..4 columns left
    } else {
      // per-lane codes; the 4 table reads and then the 4 x reads of a group are in flight
      // together, columns past the chunk's width are computed but not added
      const uint32_t* cbase = stream + (size_t)off[c] + lane;
      auto group = [&](uint32_t cwv, uint32_t j0, const bool full) {
        PatEntry e[4];
        double xs[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) e[q] = sd[(cwv >> (8u * q)) & 255u];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) xs[q] = xread(__umul24(base8, e[q].m) + e[q].off8);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const double prod = e[q].v * xs[q];
          const double sum  = acc + prod;
          acc               = ((full || j0 + q < len) && (!SKIPPAD || e[q].m != 0u)) ? sum : acc;
        }
      };
      auto pick = [&](uint32_t cwv, uint32_t j0) {
        if (j0 + 4u <= len) group(cwv, j0, true);
        else group(cwv, j0, false);
      };
      if (c == 0) {
#pragma unroll
        for (int gi = 0; gi < PF; gi++)
          if ((uint32_t)gi < ng) pick(cw[gi], (uint32_t)gi * 4u);
        for (uint32_t g = PF; g < ng; g++) pick(stream_load(cbase + (size_t)g * 64), g * 4u);
      } else {
        for (uint32_t g = 0; g < ng; g++) pick(stream_load(cbase + (size_t)g * 64), g * 4u);
      }
    }
    if (row[c] < nr) y[row[c]] = acc;
    if (DOT) {
      double t2 = row[c] < nr ? xrow[c] * acc : 0.0;
      t2        = butterfly64(t2);
      if (MASKED) pl0[c] = t2;
      else if (lane == 0) dotPartials[chunk[c]] = t2;
    }
  }
  // (MASKED) the tile's LEVEL-1 values ((q0 + q1) + q2) + q3 of its aligned groups of four chunks: the waves' level-0
  // partials meet in LDS (the 8-entry pad between the tables and the window: the masked form has no exception entries)
  // behind ONE barrier at the end of the tile's life, and the scalar step reads n/256 doubles instead of n/64 -- through
  // its single CU that was 1.9 of its 4.3 us (profiles/r03_scalar_anatomy.txt).  Same additions, same order, same bits.
  if (DOT && MASKED) {
    double* sq = reinterpret_cast<double*>(se); // 16 doubles
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < CW; c++) sq[wv + 4u * (uint32_t)c] = pl0[c];
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)CW) {
      const uint32_t gq = threadIdx.x, group = tile * (uint32_t)CW + gq;
      if (group < ((nChunks + 3u) >> 2)) dotPartials[group] = ((sq[4u * gq] + sq[4u * gq + 1u]) + sq[4u * gq + 2u]) + sq[4u * gq + 3u];
    }
  }
}


// =============================================================================
// SpMV of the CG loop WITH the p update inside (round 3): Ap = A p_new, p_new = r + beta p_old (src/CGSolver.c:114 and :123),
// and the x update the previous body owes (:127) -- the one vector kernel whose scalar is known before the SpMV starts.
// A tile forms p_new for its x window while staging it (the window's r and p_old entries come out of L2, ~4.5x redundantly
// over the tiles), stores p_new and the updated x for its OWN rows, and multiplies as spmv_scs64_pat<..., MASKED> does.  p is
// double-buffered (other tiles still read p_old); on several ranks the halo entries of the window are the neighbours' p_new
// values out of the staging area (their push kernels form them the same way, kernels.hip.h: halo_push_block<FUSEP>).
// Element for element the arithmetic is cg_update_p's (a + beta * b, x + alpha * b; separate multiply and add) followed by
// the row programs: same bits as the two kernels it replaces.
// Only for matrices whose chunks are ALL row programs (no per-lane code words: mDict == 0) and whose windows are all of the
// mapped (sigma > 1) or simple (<= 6 segments) kind; everything else keeps cg_update_p + spmv_scs64_pat.
// Register budget (one workgroup = 256 threads must stay at 8 workgroups per CU: <= 64 VGPRs): the window is staged in TWO
// halves of <= 8 slots per thread, each loading p_old AND r (16 + 16 registers) -- one dependent round trip more per tile than
// the plain kernel, instead of 30 more live registers; no code-word prefetch, no class table, no exception entries.
// =============================================================================
// (amdgpu_num_sgpr: the tile kernels must keep 8 workgroups per CU, and on this stack a wave's scalar registers are allocated
//  in 16s with 16 more on top (trap handler): above 80 the 800 per SIMD hold only 7 waves -- measured: 1792 = 7 x 256 tiles in
//  flight at 87 SGPRs, tools/make_prof_lab_fusep.py; the compiler's own occupancy model allows ~100 and fills them freely)
template <int CPT, bool SKIPPAD, bool HALO, bool MAPPED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(72))) void spmv_prog_fusep(const uint32_t* __restrict__ hdrWords, const int16_t* __restrict__ rowBase,
    const ProgBlock* __restrict__ progs, const uint16_t* __restrict__ slotMap, uint32_t mapStride,
    const double* __restrict__ pold, const double* __restrict__ r, double* __restrict__ pnew, double* xsol,
    double* __restrict__ y, const CgScalars* __restrict__ S, int which, uint32_t nr, uint32_t nChunks, uint32_t firstHdr,
    uint32_t nHdrs, uint32_t blocksPerXcd, uint32_t padCol, double* __restrict__ dotL1, HaloWait hw)
{
  extern __shared__ __attribute__((aligned(16))) double lds[]; // [16 doubles: level-1 combine][window]
  double* sq = lds;
  double* sx = lds + 16;
  constexpr int CW   = CPT / 4;
  constexpr int LONG = CPT == 8 ? 4 : 3;
  constexpr int WB   = 3 * LONG + 3;
  constexpr int H1   = 8; // slots per thread in the first half (loaded in front of the exit test)
  if (HALO && blockIdx.x < hw.nPush) { // (uniform per workgroup) this workgroup carries the rank's halo push
    if (S->stop) {
      if (blockIdx.x == 0 && threadIdx.x == 0 && halo_rank_failed(*hw.push)) halo_poison_flags(*hw.push); // this rank has failed
      return;
    }
    halo_push_block<true>(*hw.push, pold, hw.seq, blockIdx.x, hw.nPush, r, which ? 0.0 : S->beta);
    return;
  }
  const uint32_t bid   = HALO ? blockIdx.x - hw.nPush : blockIdx.x;
  const uint32_t tile0 = blocksPerXcd ? xcd_block(bid, blocksPerXcd) : bid;
  const uint32_t hidx  = firstHdr + min(tile0, nHdrs - 1u);
  const uint32_t lane  = threadIdx.x & 63u;
  uint32_t hvx, hvy = 0u;
  if (CPT == 4) {
    const uint32_t* hp = hdrWords + (size_t)hidx * 48u;
    hvx = *(lane < (uint32_t)PAT_STOP_LANE ? hp + lane : reinterpret_cast<const uint32_t*>(&S->stop));
  } else {
    const u32x2* hp = reinterpret_cast<const u32x2*>(hdrWords + (size_t)hidx * 128u);
    const u32x2 h2  = *(lane < (uint32_t)PAT_STOP_LANE ? hp + lane : reinterpret_cast<const u32x2*>(&S->stop));
    hvx = h2.x, hvy = h2.y;
  }
  // slot k of this thread: which column it holds (mapped windows).  The map's index (uniform, a 16 KB table) and the map's loads
  // (distinct maps stored once: 31 KB at HPCG 128^3, sbhip_matrix.inc.h) leave together with the header's load, so the map is
  // there when the header is: 28.5-28.6 against 28.7-29.0 us per launch, and 53 instead of 63 VGPRs (two builds, same box,
  // alternating).  (With one map per tile -- 21 MB per launch out of the Infinity Cache -- the early fetch cost registers and
  // bought nothing: rounds 2-3.)
  // (the second half's map entries ride in the high halves of the first half's registers: 8 live registers instead of 15)
  uint32_t dmapE[MAPPED ? H1 : 1];
  if (MAPPED) {
    const uint32_t mStr = mapStride & 0xFFFu, mOff = (mapStride >> 12) * 256u;
    const uint32_t mIdx = mOff ? reinterpret_cast<const uint32_t*>(slotMap)[hidx] : hidx;
    const uint16_t* mp  = slotMap + mOff + (size_t)mIdx * mStr + threadIdx.x;
    uint32_t lo[H1], hi[H1];
#pragma unroll
    for (int k = 0; k < H1; k++) {
      lo[k] = mp[min((uint32_t)k * 256u, mStr - 256u)];
      hi[k] = k + H1 < WB ? (uint32_t)mp[min((uint32_t)(k + H1) * 256u, mStr - 256u)] : 0u;
    }
#pragma unroll
    for (int k = 0; k < H1; k++) dmapE[k] = lo[k] | (hi[k] << 16);
  }
  // the step's scalars (uniform: scalar loads, back with the header)
  const double beta  = which ? 0.0 : S->beta; // k = 1: p = r + 0.0 * r (:109), the host passes pold = r
  const double alpha = S->alpha;
  const bool owed    = !which && S->x_pending != 0;
  auto field = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)hvx, i); };
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto cfield = [&](int c, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)(c == 0 ? hvx : hvy), i); };
  const int stopped   = (int)field(PAT_STOP_LANE);
  const uint32_t tile = field(46), flags = field(31), win = field(3);
  if (HALO && (flags & PAT_TOUCHES_HALO) && !stopped && tile0 < nHdrs) { // wait for the neighbours' blocks
    if ((int)threadIdx.x < hw.nsrc) {
      const unsigned long long* f = hw.flags + (hw.seq & 1ull) * P2P_MAX + hw.src[threadIdx.x];
      const int how               = p2p_wait(f, hw.seq, hw.timeoutTicks);
      if (how) halo_wait_failed(how, hw.err, hw.stopw);
    }
    __syncthreads();
  }
  // where slot k goes, whether it exists
  uint32_t dmap[MAPPED ? H1 : 1];
#pragma unroll
  for (int k = 0; k < (MAPPED ? H1 : 1); k++) dmap[k] = dmapE[k];
  auto col_of = [&](int k) -> uint32_t {
    if (MAPPED) return field(12 + min(k, 17)) + (k < H1 ? dmap[MAPPED ? k : 0] & 0xFFFFu : dmap[MAPPED ? k - H1 : 0] >> 16);
    const int sI = k < 3 * LONG ? k / LONG : 3 + (k - 3 * LONG);
    const uint32_t i = (k < 3 * LONG ? (uint32_t)(k % LONG) * 256u : 0u) + threadIdx.x;
    const uint32_t sc = field(12 + 3 * sI), sn = field(12 + 3 * sI + 2);
    return sn ? sc + min(i, sn - 1u) : padCol;
  };
  auto slot_of = [&](int k, bool& valid) -> uint32_t {
    if (MAPPED) {
      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;
      valid = slot < win;
      return slot;
    }
    const int sI = k < 3 * LONG ? k / LONG : 3 + (k - 3 * LONG);
    const uint32_t i = (k < 3 * LONG ? (uint32_t)(k % LONG) * 256u : 0u) + threadIdx.x;
    valid = i < field(12 + 3 * sI + 2);
    return field(12 + 3 * sI + 1) + i;
  };
  // p_new of a window column: own columns r + beta p_old, halo columns the neighbour's p_new out of the staging area
  double pv[H1], rv[H1];
  uint32_t isHalo = 0u;
#pragma unroll
  for (int k = 0; k < H1; k++) {
    const uint32_t c = col_of(k);
    if (HALO) {
      const bool h = c >= nr;
      isHalo |= (h ? 1u : 0u) << k;
      pv[k] = *(h ? hw.ext + (c - nr) : pold + c);
      rv[k] = r[h ? 0u : c];
    } else {
      pv[k] = pold[c], rv[k] = r[c];
    }
  }
  uint32_t chunk[CW], row[CW], lenf[CW];
  int32_t base[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    chunk[c] = tile * CPT + wv + 4u * (uint32_t)c;
    row[c]   = chunk[c] * 64u + lane;
    lenf[c]  = cfield(c, 8 + (int)wv);
    base[c]  = (int32_t)rowBase[chunk[c] < nChunks ? row[c] : 0u];
  }
#pragma unroll
  for (int k = 0; k < H1; k++) asm volatile("" ::"v"(pv[k]), "v"(rv[k]));
#pragma unroll
  for (int c = 0; c < CW; c++) asm volatile("" ::"v"(base[c]));
  if (tile0 >= nHdrs || stopped) return; // uniform per workgroup
  // slot 0: what padding multiplies, p_new[padCol] (uniform addresses: scalar loads; formed again where a chunk needs it)
  if (!MAPPED && threadIdx.x == 0) sx[0] = r[padCol] + beta * pold[padCol];
#pragma unroll
  for (int k = 0; k < H1; k++) {
    bool valid;
    const uint32_t slot = slot_of(k, valid);
    const double pn     = (HALO && ((isHalo >> k) & 1u)) ? pv[k] : rv[k] + beta * pv[k];
    if (valid) sx[slot] = pn;
  }
  { // second half: the same registers again (one more dependent round trip instead of 30 more live registers)
    double pw[WB - H1], rw[WB - H1];
    uint32_t isHalo2 = 0u;
#pragma unroll
    for (int k = H1; k < WB; k++) {
      const uint32_t c = col_of(k);
      if (HALO) {
        const bool h = c >= nr;
        isHalo2 |= (h ? 1u : 0u) << (k - H1);
        pw[k - H1] = *(h ? hw.ext + (c - nr) : pold + c);
        rw[k - H1] = r[h ? 0u : c];
      } else {
        pw[k - H1] = pold[c], rw[k - H1] = r[c];
      }
    }
#pragma unroll
    for (int k = H1; k < WB; k++) {
      bool valid;
      const uint32_t slot = slot_of(k, valid);
      const double pn     = (HALO && ((isHalo2 >> (k - H1)) & 1u)) ? pw[k - H1] : rw[k - H1] + beta * pw[k - H1];
      if (valid) sx[slot] = pn;
    }
  }
  if (MAPPED && threadIdx.x == 0) sx[0] = r[padCol] + beta * pold[padCol]; // (behind this thread's own store to slot 0)
  __syncthreads();
  // own rows: p_old, r and (if the previous body owes it) x, behind the barrier so that their latency hides behind the programs
  double rown[CW], xown[CW], xrow[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const uint32_t rr = min(row[c], nr - 1u);
    xrow[c] = pold[rr]; // x update, own p_new, dot
    rown[c] = r[rr];
    xown[c] = owed ? xsol[rr] : 0.0;
  }
  constexpr uint32_t sxOff = 16u * (uint32_t)sizeof(double);
  double pl0[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) pl0[c] = 0.0;
#pragma unroll
  for (int c = 0; c < CW; c++) {
    if (chunk[c] >= nChunks) continue; // wave-uniform
    const uint32_t len    = lenf[c] & PAT_LEN_MASK;
    double acc            = 0.0;
    const uint32_t base8x = ((uint32_t)base[c] << 3) + sxOff;
    const ProgBlock* pg   = progs + cfield(c, 32 + (int)wv);
    const bool nomask     = (lenf[c] & PAT_NOPAD) != 0u; // wave-uniform: every lane has every entry
    const char* ldsBytes  = reinterpret_cast<const char*>(lds);
    uint32_t j0           = 0;
    if (nomask) {
#pragma unroll 1
      for (; j0 + 8u <= len; j0 += 8u) prog_batch<8, false>(pg + (j0 >> 3), base8x, ldsBytes, acc);
    } else {
#pragma unroll 1
      for (; j0 + 8u <= len; j0 += 8u) prog_batch<8, true>(pg + (j0 >> 3), base8x, ldsBytes, acc);
    }
    if (j0 + 4u < len) prog_batch<8, true>(pg + (j0 >> 3), base8x, ldsBytes, acc);
    else if (j0 < len) prog_batch<4, true>(pg + (j0 >> 3), base8x, ldsBytes, acc);
    if (!SKIPPAD && (lenf[c] & PAT_HASPAD)) {
      const unsigned long long padMask =
          (unsigned long long)cfield(c, 36 + 2 * (int)wv) | ((unsigned long long)cfield(c, 37 + 2 * (int)wv) << 32);
      const double xpad = r[padCol] + beta * pold[padCol]; // (== sx[0]; the same two scalar loads and operations)
      masked_add(acc, 0.0 * xpad, padMask);
    }
    double t2 = 0.0;
    if (row[c] < nr) {
      y[row[c]] = acc;
      const double po = xrow[c], pn = rown[c] + beta * po; // cg_update_p's arithmetic for the own row
      pnew[row[c]] = pn;
      if (owed) xsol[row[c]] = xown[c] + alpha * po;
      t2 = pn * acc;
    }
    pl0[c] = butterfly64(t2);
  }
  // the tile's level-1 values of p_new . Ap (as spmv_scs64_pat<..., MASKED> forms them)
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < CW; c++) sq[wv + 4u * (uint32_t)c] = pl0[c];
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)CW) {
    const uint32_t gq = threadIdx.x, group = tile * (uint32_t)CW + gq;
    if (group < ((nChunks + 3u) >> 2)) dotL1[group] = ((sq[4u * gq] + sq[4u * gq + 1u]) + sq[4u * gq + 2u]) + sq[4u * gq + 3u];
  }
}

} // namespace sbk
