// kernels.hip.h -- hand-written CDNA4 (gfx950, wave64) kernels of the CG hot path.
//
// All arithmetic is IEEE fp64 with separate multiply and add (the TU is compiled
// with -ffp-contract=off): per element the results are bit-identical to the
// reference's strict-IEEE CPU loops, and dot products use ONE fixed summation
// order (DESIGN.md "dot order") so they are reproducible and can be restated on
// the CPU (oracle/sb_oracle.c: orc_ddot_partials / orc_reduce_final).
//
// Everything here is HBM-bandwidth bound (0.16 flop/byte): no MFMA; the levers
// are coalesced 512 B..1 KiB wave-instructions, enough loads in flight per CU,
// keeping the gathered vector in the XCD-local L2, and fusing reductions into the
// kernel that already holds the operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbk {

constexpr int WAVE = 64;

// Control block shared by the CG kernels (lives in HBM, never read by the host
// inside the loop).
struct CgScalars {
  double rr;      // rtrans          (src/CGSolver.c:83)
  double rr_old;  // oldrtrans
  double pAp;
  double alpha;   // rtrans / pAp    (:126)
  double beta;    // rtrans / oldrtrans (:113)
  double neg_alpha;
  double local;   // rank-local sum handed to the all-reduce
  double pad;
  int stop;       // 1: the reference's loop has exited; every kernel returns
  int stop_next;  // !(normr > eps) as of the last r.r (loop condition, :107)
  int iters;      // last k whose body ran
  int n_rr;       // entries in rr_hist
  int n_pAp;
  int pad2[3];
};

// ---- wave-level fixed-order reductions --------------------------------------
// xor butterfly, offsets 1,2,4,...: every lane ends with the same value because
// fp add is commutative.  This IS the level-0 order of the canonical dot.
__device__ __forceinline__ double butterfly64(double v)
{
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
// half-wave form: lanes 0-31 and 32-63 each reduce their own 32 values
__device__ __forceinline__ double butterfly32(double v)
{
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its
// 4 MiB L2).  Give each XCD one contiguous range of logical blocks so that the
// slice of x its chunks gather stays in ITS L2.  Speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t per_xcd)
{
  return (b & 7u) * per_xcd + (b >> 3);
}

template <typename T> __device__ __forceinline__ T stream_load(const T* p)
{
  return __builtin_nontemporal_load(p); // matrix data is read exactly once per SpMV
}

// =============================================================================
// Sell-C-sigma SpMV, C = 64: one wavefront per chunk, lane k = row k of the chunk
// (reference loop: src/matrix-SCS.c:208-227, its inner k loop is our lane axis).
// val/colInd are column-major inside the chunk, so each wave-instruction reads
// 512 B of val and 256 B of colInd, fully coalesced.  Each lane accumulates its
// row left to right exactly like the CPU loop.  DOT fuses p.Ap (level 0 of the
// canonical order: chunk i == 64-group i of the output vector).
// =============================================================================
template <int UNROLL, bool DOT, bool NT>
__global__ __launch_bounds__(256) void spmv_scs64(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
    uint32_t nr, uint32_t nChunks, uint32_t blocksPerXcd, double* __restrict__ dotPartials,
    const int* __restrict__ stop)
{
  const int stopped    = stop ? *stop : 0; // one wait covers this and the loads below
  const uint32_t lb    = blocksPerXcd ? xcd_block(blockIdx.x, blocksPerXcd) : blockIdx.x;
  const uint32_t chunk = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const uint32_t cp  = chunkPtr[chunk];
  const uint32_t len = chunkLens[chunk];
  if (stopped) return;
  const double* v    = val + cp + lane;
  const uint32_t* c  = colInd + cp + lane;
  double acc         = 0.0;
  uint32_t j         = 0;
  for (; j + UNROLL <= len; j += UNROLL) {
    double vv[UNROLL];
    uint32_t cc[UNROLL];
    double xx[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      vv[u] = NT ? stream_load(v + (size_t)(j + u) * 64) : v[(size_t)(j + u) * 64];
      cc[u] = NT ? stream_load(c + (size_t)(j + u) * 64) : c[(size_t)(j + u) * 64];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) xx[u] = x[cc[u]];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc = acc + vv[u] * xx[u];
  }
  for (; j < len; j++) {
    double vv   = NT ? stream_load(v + (size_t)j * 64) : v[(size_t)j * 64];
    uint32_t cc = NT ? stream_load(c + (size_t)j * 64) : c[(size_t)j * 64];
    acc         = acc + vv * x[cc];
  }
  const uint32_t row = chunk * 64u + lane;
  if (row < nr) y[row] = acc;
  if (DOT) {
    double t = row < nr ? x[row] * acc : 0.0;
    t        = butterfly64(t);
    if (lane == 0) dotPartials[chunk] = t;
  }
}

// Software-pipelined form of the same kernel.  A wave streams its chunk in batches of
// U columns; the val/colInd loads of batch b+1 are issued right after the x-gathers
// of batch b, so the HBM latency of the stream overlaps the L2 latency of the gather
// instead of adding to it.  Loads may run up to U-1 columns past the chunk's end (the
// arrays carry SCS_SLACK elements of zero padding, and a following chunk's indices
// are valid columns); such columns are never accumulated.  Same per-row order, same
// bits as spmv_scs64.
constexpr uint32_t SCS_SLACK = 16 * 64;

template <int U, bool DOT>
__global__ __launch_bounds__(256) void spmv_scs64_pipe(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
    uint32_t nr, uint32_t nChunks, uint32_t blocksPerXcd, double* __restrict__ dotPartials,
    const int* __restrict__ stop)
{
  const int stopped    = stop ? *stop : 0; // issued together with the loads below
  const uint32_t lb    = blocksPerXcd ? xcd_block(blockIdx.x, blocksPerXcd) : blockIdx.x;
  const uint32_t chunk = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
  const uint32_t lane  = threadIdx.x & 63u;
  if (chunk >= nChunks) return;
  const uint32_t cp  = chunkPtr[chunk];
  const uint32_t len = chunkLens[chunk];
  if (stopped) return;
  const double* v   = val + cp + lane;
  const uint32_t* c = colInd + cp + lane;
  double acc        = 0.0;
  double va[U], vb[U];
  uint32_t ca[U], cb[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    va[u] = stream_load(v + (size_t)u * 64);
    ca[u] = stream_load(c + (size_t)u * 64);
  }
  for (uint32_t j = 0; j < len; j += U) {
    double xx[U];
#pragma unroll
    for (int u = 0; u < U; u++) xx[u] = x[ca[u]];
    if (j + U < len) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        vb[u] = stream_load(v + (size_t)(j + U + u) * 64);
        cb[u] = stream_load(c + (size_t)(j + U + u) * 64);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (j + u < len) acc = acc + va[u] * xx[u];
#pragma unroll
    for (int u = 0; u < U; u++) va[u] = vb[u], ca[u] = cb[u];
  }
  const uint32_t row = chunk * 64u + lane;
  if (row < nr) y[row] = acc;
  if (DOT) {
    double t = row < nr ? x[row] * acc : 0.0;
    t        = butterfly64(t);
    if (lane == 0) dotPartials[chunk] = t;
  }
}

// Any C (the reference's fixtures use C = 1, 2, 4): one thread per padded row.
// Coalesced whenever C is a multiple of 64; correctness path otherwise.
__global__ __launch_bounds__(256) void spmv_scs_generic(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
    uint32_t nr, uint32_t nrPadded, uint32_t C, const int* __restrict__ stop)
{
  if (stop && *stop) return;
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrPadded) return;
  const uint32_t chunk = row / C;
  const uint32_t k     = row - chunk * C;
  const uint32_t cp    = chunkPtr[chunk];
  const uint32_t len   = chunkLens[chunk];
  double acc           = 0.0;
  for (uint32_t j = 0; j < len; j++) {
    const size_t idx = (size_t)cp + (size_t)j * C + k;
    acc              = acc + val[idx] * x[colInd[idx]];
  }
  if (row < nr) y[row] = acc;
}

// =============================================================================
// CRS SpMV (reference loop: src/matrix-CRS.c:54-64).  Row blocks are cut on the
// host so that a block's nonzeros fit one LDS tile: the workgroup streams
// val*x[col] products into LDS with coalesced loads, then thread r adds row r's
// products left to right -- the CPU's order, hence the CPU's bits -- from LDS.
// A row longer than the tile is walked tile by tile by thread 0 (still in order).
// =============================================================================
constexpr int CRS_THREADS = 256;
constexpr int CRS_TILE    = 2048; // nonzeros per LDS tile (16 KiB: 8 workgroups per CU)
constexpr int CRS_BATCH   = CRS_TILE / CRS_THREADS; // products per thread per tile

// products of nonzeros [base, end) -> prod[]: all of a thread's loads are issued before
// the first use (stream loads, then gathers), one HBM + one cache round trip per tile
__device__ __forceinline__ void crs_products(uint32_t base, uint32_t end, uint32_t t,
    const uint32_t* __restrict__ colInd, const double* __restrict__ val, const double* __restrict__ x,
    double* prod)
{
  double v[CRS_BATCH], xv[CRS_BATCH];
  uint32_t c[CRS_BATCH];
#pragma unroll
  for (int u = 0; u < CRS_BATCH; u++) {
    const uint32_t k = base + t + (uint32_t)u * CRS_THREADS;
    v[u] = 0.0, c[u] = 0u;
    if (k < end) v[u] = stream_load(val + k), c[u] = stream_load(colInd + k);
  }
#pragma unroll
  for (int u = 0; u < CRS_BATCH; u++) {
    const uint32_t k = base + t + (uint32_t)u * CRS_THREADS;
    xv[u]            = k < end ? x[c[u]] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < CRS_BATCH; u++) {
    const uint32_t k = base + t + (uint32_t)u * CRS_THREADS;
    if (k < end) prod[k - base] = v[u] * xv[u];
  }
}

__global__ __launch_bounds__(CRS_THREADS) void spmv_crs_stream(
    const uint32_t* __restrict__ rowBlocks, const uint32_t* __restrict__ rowPtr,
    const uint32_t* __restrict__ colInd, const double* __restrict__ val,
    const double* __restrict__ x, double* __restrict__ y, uint32_t nBlocks,
    uint32_t blocksPerXcd, const int* __restrict__ stop)
{
  __shared__ double prod[CRS_TILE];
  const int stopped = stop ? *stop : 0;
  const uint32_t lb = xcd_block(blockIdx.x, blocksPerXcd);
  if (lb >= nBlocks || stopped) return;
  const uint32_t r0 = rowBlocks[lb], r1 = rowBlocks[lb + 1];
  const uint32_t n0 = rowPtr[r0], n1 = rowPtr[r1];
  const uint32_t t = threadIdx.x;
  if (n1 - n0 <= (uint32_t)CRS_TILE) {
    const uint32_t r = r0 + t; // this thread's row (if any): fetch its extent early
    uint32_t a = 0, b = 0;
    if (r < r1) a = rowPtr[r] - n0, b = rowPtr[r + 1] - n0;
    crs_products(n0, n1, t, colInd, val, x, prod);
    __syncthreads();
    if (r < r1) {
      double sum = 0.0;
      for (uint32_t k = a; k < b; k++) sum = sum + prod[k];
      y[r] = sum;
    }
  } else { // one very long row (the host never puts two rows in an oversize block)
    double sum = 0.0;
    for (uint32_t base = n0; base < n1; base += CRS_TILE) {
      const uint32_t end = min(base + (uint32_t)CRS_TILE, n1);
      __syncthreads();
      crs_products(base, end, t, colInd, val, x, prod);
      __syncthreads();
      if (t == 0)
        for (uint32_t k = 0; k < end - base; k++) sum = sum + prod[k];
    }
    if (t == 0) y[r0] = sum;
  }
}

// =============================================================================
// BLAS-1
// =============================================================================
// waxpby (src/solver.c:16-39).  Under strict IEEE the reference's three branches
// are bitwise equal to alpha*x + beta*y (1.0*x is exact), so one form serves.
// 16 B per lane; w may alias x or y (each element is read before it is written
// by the same lane).
__global__ __launch_bounds__(256) void waxpby_k(uint32_t n, double alpha, const double* x,
    double beta, const double* y, double* w, const int* __restrict__ stop)
{
  if (stop && *stop) return;
  const uint32_t n2     = n >> 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const double2* x2     = reinterpret_cast<const double2*>(x);
  const double2* y2     = reinterpret_cast<const double2*>(y);
  double2* w2           = reinterpret_cast<double2*>(w);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 a = x2[i], b = y2[i], r;
    r.x = alpha * a.x + beta * b.x;
    r.y = alpha * a.y + beta * b.y;
    w2[i] = r;
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0)
    w[n - 1] = alpha * x[n - 1] + beta * y[n - 1];
}

// waxpby with the scalar of y taken from HBM: w = x + (*beta)*y.  This is how the
// reference-shaped (unfused) CG issues "x = x + alpha p" and "r = r - alpha Ap"
// (src/CGSolver.c:127-128) without a host round trip for alpha.
__global__ __launch_bounds__(256) void waxpby_sdev_k(uint32_t n, const double* x,
    const double* __restrict__ beta_dev, const double* y, double* w,
    const int* __restrict__ stop)
{
  if (stop && *stop) return;
  const double beta     = *beta_dev;
  const uint32_t n2     = n >> 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const double2* x2     = reinterpret_cast<const double2*>(x);
  const double2* y2     = reinterpret_cast<const double2*>(y);
  double2* w2           = reinterpret_cast<double2*>(w);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 a = x2[i], b = y2[i], r;
    r.x = a.x + beta * b.x;
    r.y = a.y + beta * b.y;
    w2[i] = r;
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) w[n - 1] = x[n - 1] + beta * y[n - 1];
}

// same, scalars read from the device-resident control block (no host round trip):
// which = 0: p = r + beta*p            (src/CGSolver.c:114)
// which = 1: p = r + 0.0*r             (:109, the literal k==1 form)
__global__ __launch_bounds__(256) void cg_update_p(uint32_t n, const double* __restrict__ r,
    double* p, const CgScalars* __restrict__ S, int which)
{
  if (S->stop) return;
  const double beta     = which == 0 ? S->beta : 0.0;
  const uint32_t n2     = n >> 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const double2* r2     = reinterpret_cast<const double2*>(r);
  double2* p2           = reinterpret_cast<double2*>(p);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 a = r2[i], b = which == 0 ? p2[i] : a, o;
    o.x = a.x + beta * b.x;
    o.y = a.y + beta * b.y;
    p2[i] = o;
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double b = which == 0 ? p[n - 1] : r[n - 1];
    p[n - 1]       = r[n - 1] + beta * b;
  }
}

// Level 0 of the canonical dot: partials[g] = butterfly over elements 64g..64g+63.
// A wave covers 128 elements per step with 16-B loads: lane l holds elements
// 2l, 2l+1; its in-lane add is butterfly offset 1, lane-xor 1..16 are offsets
// 2..32; lanes 0-31 own group 2s, lanes 32-63 group 2s+1.
__device__ __forceinline__ void dot_span(uint32_t span, uint32_t n, const double* x,
    const double* y, double* partials, uint32_t lane)
{
  const uint32_t e = span * 128u + lane * 2u;
  double t         = 0.0;
  if (e + 1 < n) {
    const double2 a = *reinterpret_cast<const double2*>(x + e);
    const double2 b = *reinterpret_cast<const double2*>(y + e);
    t               = a.x * b.x + a.y * b.y;
  } else if (e < n) {
    t = x[e] * y[e] + 0.0;
  }
  t = butterfly32(t);
  const uint32_t g = span * 2u + (lane >> 5);
  if ((lane & 31u) == 0 && g * 64u < n) partials[g] = t;
}

__global__ __launch_bounds__(256) void ddot_partials_k(uint32_t n, const double* x,
    const double* y, double* __restrict__ partials, const int* __restrict__ stop)
{
  if (stop && *stop) return;
  const uint32_t lane   = threadIdx.x & 63u;
  const uint32_t nSpans = (n + 127u) >> 7;
  const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nSpans; s += nWaves)
    dot_span(s, n, x, y, partials, lane);
}

// x += alpha p ; r -= alpha Ap ; partials of the NEW r.r  (src/CGSolver.c:127-128
// and the ddot of :112 of the next iteration) in one pass: 4 streams in, 2 out.
__global__ __launch_bounds__(256) void cg_update_xr_dot(uint32_t n, double* x,
    const double* __restrict__ p, double* r, const double* __restrict__ Ap,
    const CgScalars* __restrict__ S, double* __restrict__ partials)
{
  if (S->stop) return;
  const double alpha    = S->alpha;
  const double nalpha   = -alpha;
  const uint32_t lane   = threadIdx.x & 63u;
  const uint32_t nSpans = (n + 127u) >> 7;
  const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nSpans; s += nWaves) {
    const uint32_t e = s * 128u + lane * 2u;
    double t         = 0.0;
    if (e + 1 < n) {
      double2 xv        = *reinterpret_cast<double2*>(x + e);
      const double2 pv  = *reinterpret_cast<const double2*>(p + e);
      double2 rv        = *reinterpret_cast<double2*>(r + e);
      const double2 av  = *reinterpret_cast<const double2*>(Ap + e);
      xv.x = xv.x + alpha * pv.x;
      xv.y = xv.y + alpha * pv.y;
      rv.x = rv.x + nalpha * av.x;
      rv.y = rv.y + nalpha * av.y;
      *reinterpret_cast<double2*>(x + e) = xv;
      *reinterpret_cast<double2*>(r + e) = rv;
      t = rv.x * rv.x + rv.y * rv.y;
    } else if (e < n) {
      x[e]            = x[e] + alpha * p[e];
      const double rn = r[e] + nalpha * Ap[e];
      r[e]            = rn;
      t               = rn * rn + 0.0;
    }
    t = butterfly32(t);
    const uint32_t g = s * 2u + (lane >> 5);
    if ((lane & 31u) == 0 && g * 64u < n) partials[g] = t;
  }
}

// r = b - Ap with the r.r partials fused (prologue, src/CGSolver.c:97-98)
__global__ __launch_bounds__(256) void cg_residual_dot(uint32_t n, const double* __restrict__ b,
    const double* __restrict__ Ap, double* __restrict__ r, double* __restrict__ partials)
{
  const uint32_t lane   = threadIdx.x & 63u;
  const uint32_t nSpans = (n + 127u) >> 7;
  const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nSpans; s += nWaves) {
    const uint32_t e = s * 128u + lane * 2u;
    double t         = 0.0;
    if (e + 1 < n) {
      const double2 bv = *reinterpret_cast<const double2*>(b + e);
      const double2 av = *reinterpret_cast<const double2*>(Ap + e);
      double2 rv;
      rv.x = bv.x + -1.0 * av.x;
      rv.y = bv.y + -1.0 * av.y;
      *reinterpret_cast<double2*>(r + e) = rv;
      t = rv.x * rv.x + rv.y * rv.y;
    } else if (e < n) {
      const double rn = b[e] + -1.0 * Ap[e];
      r[e]            = rn;
      t               = rn * rn + 0.0;
    }
    t = butterfly32(t);
    const uint32_t g = s * 2u + (lane >> 5);
    if ((lane & 31u) == 0 && g * 64u < n) partials[g] = t;
  }
}

// Level 1 of the canonical dot: ONE workgroup of 1024 threads.  Thread t adds
// partials t, t+1024, ... in order; each wave butterflies; the 16 wave sums are
// added in wave order by thread 0.
__device__ __forceinline__ double reduce_final_block(uint32_t m, const double* __restrict__ q,
    double* lds16)
{
  double s   = 0.0;
  uint32_t i = threadIdx.x;
  // same order as the plain loop, but 32 / 8 independent loads are in flight at a time
  // (the partials were written by other CUs: every load is an L2/fabric round trip,
  // and this single workgroup sits on the critical path of every CG iteration)
  for (; i + 31u * 1024u < m; i += 32u * 1024u) {
    double a[32];
#pragma unroll
    for (int u = 0; u < 32; u++) a[u] = q[i + (uint32_t)u * 1024u];
#pragma unroll
    for (int u = 0; u < 32; u++) s = s + a[u];
  }
  for (; i + 7u * 1024u < m; i += 8u * 1024u) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = q[i + (uint32_t)u * 1024u];
#pragma unroll
    for (int u = 0; u < 8; u++) s = s + a[u];
  }
  for (; i < m; i += 1024u) s = s + q[i];
  s = butterfly64(s);
  if ((threadIdx.x & 63u) == 0) lds16[threadIdx.x >> 6] = s;
  __syncthreads();
  double total = lds16[0];
#pragma unroll
  for (int w = 1; w < 16; w++) total = total + lds16[w];
  return total; // every thread returns the same value
}

__global__ __launch_bounds__(1024) void reduce_final_k(uint32_t m, const double* __restrict__ q,
    double* __restrict__ out, const int* __restrict__ stop)
{
  __shared__ double lds16[16];
  if (stop && *stop) return;
  const double total = reduce_final_block(m, q, lds16);
  if (threadIdx.x == 0) *out = total;
}

// CG scalar steps.  MODE: 0 prologue r.r, 1 loop r.r (top of iteration k >= 2),
// 2 p.Ap.  When REDUCE is false the (all-reduced) sum is already in S->local.
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(1024) void cg_scalar_k(uint32_t m, const double* __restrict__ q,
    CgScalars* S, double eps, double* __restrict__ rr_hist, double* __restrict__ pAp_hist,
    int hist_cap, int to_local_only)
{
  __shared__ double lds16[16];
  if (S->stop) return;
  if (MODE == 1 && S->stop_next) { // the reference's `normr > eps` test failed: loop exits
    __syncthreads();
    if (threadIdx.x == 0) S->stop = 1;
    return;
  }
  double total;
  if (REDUCE) {
    total = reduce_final_block(m, q, lds16);
    if (to_local_only) { // multi-rank: hand the local sum to the all-reduce
      if (threadIdx.x == 0) S->local = total;
      return;
    }
  } else {
    total = S->local;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (MODE == 0) {
    S->rr        = total;
    S->stop_next = !(sqrt(total) > eps);
    if (S->n_rr < hist_cap) rr_hist[S->n_rr] = total;
    S->n_rr++;
  } else if (MODE == 1) {
    const double old = S->rr;
    S->rr_old        = old;
    S->rr            = total;
    S->beta          = total / old;
    S->stop_next     = !(sqrt(total) > eps);
    S->iters         = S->iters + 1;
    if (S->n_rr < hist_cap) rr_hist[S->n_rr] = total;
    S->n_rr++;
  } else {
    S->pAp          = total;
    const double al = S->rr / total;
    S->alpha        = al;
    S->neg_alpha    = -al;
    if (S->n_pAp < hist_cap) pAp_hist[S->n_pAp] = total;
    S->n_pAp++;
  }
}

// top of iteration k == 1 (no r.r there): apply the loop condition
__global__ void cg_iter1_begin(CgScalars* S)
{
  if (S->stop) return;
  if (S->stop_next) S->stop = 1;
  else S->iters = 1;
}

// =============================================================================
// permutation / halo helpers
// =============================================================================
__global__ __launch_bounds__(256) void gather_k(uint32_t n, const uint32_t* __restrict__ idx,
    const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ stop)
{ // out[i] = in[idx[i]]   (halo pack: src/comm.c:635-638)
  if (stop && *stop) return;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[idx[i]];
}

__global__ __launch_bounds__(256) void scatter_k(uint32_t n, const uint32_t* __restrict__ idx,
    const double* __restrict__ in, double* __restrict__ out)
{ // out[idx[i]] = in[i]
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[idx[i]] = in[i];
}

__global__ __launch_bounds__(256) void max_abs_diff_partials(uint32_t n,
    const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out)
{ // solverCheckResidual, src/CGSolver.c:50-53 (max is order-independent)
  __shared__ double w[4];
  double m              = 0.0;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double d = fabs(a[i] - b[i]);
    if (d > m) m = d;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_xor(m, off, 64);
    if (o > m) m = o;
  }
  if ((threadIdx.x & 63u) == 0) w[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++)
      if (w[i] > m) m = w[i];
    out[blockIdx.x] = m;
  }
}

// debug: pure streaming read (16 B per lane), result folded so nothing is elided
__global__ __launch_bounds__(256) void stream_read_k(const double2* __restrict__ in, size_t n2,
    double* __restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc          = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 v = in[i];
    acc += v.x + v.y;
  }
  if (acc == 123.456) out[0] = acc;
}

} // namespace sbk
