// kernels.hip.h -- hand-written CDNA4 (gfx950, wave64) kernels of the CG hot path.
//
// All arithmetic is IEEE fp64 with separate multiply and add (the TU is compiled
// with -ffp-contract=off): per element the results are bit-identical to the
// reference's strict-IEEE CPU loops, and dot products use ONE fixed summation
// order (DESIGN.md "dot order") so they are reproducible and can be restated on
// the CPU by the test oracle.
//
// Nothing here is a dense contraction (0.16 flop/byte): no MFMA.  The levers are
// coalesced 256 B..1 KiB wave-instructions, bytes in flight per CU, x kept in the
// XCD-local L2 / in LDS, few bytes per nonzero (pack.hip.h), and few launches: dots are
// fused into the kernels that already hold their operands.  (Finishing a reduction INSIDE
// its producer -- last-arriving workgroup, sharded tickets -- was built and measured: every
// workgroup then ends on a store drain + returning atomic, SpMV 50 -> 71 us for 12 us of
// launches saved.  Removed; see DESIGN.md.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbk {

// Control block of the CG loop (lives in HBM; the host writes it once per solve and
// reads it once at the end).
struct CgScalars {
  double rr;      // rtrans          (src/CGSolver.c:83)
  double rr_old;  // oldrtrans
  double pAp;
  double alpha;   // rtrans / pAp    (:126)
  double beta;    // rtrans / oldrtrans (:113)
  double neg_alpha;
  double local;   // rank-local sum handed to / returned by the all-reduce
  double eps;
  int stop;       // 1: the reference's for loop has exited; every kernel returns
  int stop_next;  // !(normr > eps) for the normr the NEXT loop test will see (:107,:116)
  int iters;      // k of the last loop body that runs / ran
  int n_rr;       // entries written to rr_hist
  int n_pAp;
  int itermax;
  int hist_cap;
  int x_pending;  // 1: x += alpha p of the last body is still owed (applied by the next p update)
  int p2p_error;  // 1: a peer's contribution to an in-kernel all-reduce did not arrive in time
  // what the NEXT beta step will read of this block, copied by every alpha step (cg_apply<2>): where the beta step rides at
  // the head of the p update (cg_update_p<BETA>) every workgroup takes it from these fields while workgroup 0 is already
  // overwriting rr / iters / stop_next with the step's results -- a late workgroup must not see the new values
  double snap_rr;
  int snap_iters, snap_stop_next;
};

// ---- wave-level fixed-order reductions --------------------------------------
// xor butterfly, offsets 1,2,4,...: every lane ends with the same value because
// fp add is commutative.  This IS level 0 of the canonical dot.
// The partner's value arrives by DPP moves (offsets 1, 2: quad_perm; 4: row_shl / row_shr by bank; 8: row_ror) and by
// gfx950's v_permlane16_swap / v_permlane32_swap (offsets 16, 32: A' + B' of the swapped pair is own + partner in every
// lane), not by __shfl_xor, which compiles to two ds_bpermute per step: 12 dependent LDS-pipe round trips per 64-bit
// butterfly, 223 ns against 92 ns (tools/lab/butterfly_lab.hip, which also checks every step bit for bit) -- and in the
// SpMV they sit at the very end of a tile's life, where nothing hides them.
template <int CTRL, int BANKMASK> __device__ __forceinline__ double dpp_move(double old, double v)
{ // lanes of the banks in BANKMASK: v of the lane CTRL names; the others: old
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v), o = __builtin_bit_cast(unsigned long long, old);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)b, CTRL, 0xF, BANKMASK, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(b >> 32), CTRL, 0xF, BANKMASK, false);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <int W> __device__ __forceinline__ double swap_add(double v)
{ // v + (v of lane ^ W), W = 16 or 32
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const auto lo = W == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false)
                          : __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
  const auto hi = W == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false)
                          : __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  const double A = __builtin_bit_cast(double, (unsigned long long)lo[0] | ((unsigned long long)hi[0] << 32));
  const double B = __builtin_bit_cast(double, (unsigned long long)lo[1] | ((unsigned long long)hi[1] << 32));
  return A + B;
}
__device__ __forceinline__ double butterfly16(double v)
{ // offsets 1, 2, 4, 8
  v = v + dpp_move<0xB1, 0xF>(v, v); // quad_perm [1,0,3,2]
  v = v + dpp_move<0x4E, 0xF>(v, v); // quad_perm [2,3,0,1]
  v = v + dpp_move<0x114, 0xA>(dpp_move<0x104, 0x5>(v, v), v); // banks 0,2: row_shl:4 (lane + 4); banks 1,3: row_shr:4 (lane - 4)
  return v + dpp_move<0x128, 0xF>(v, v); // row_ror:8
}
__device__ __forceinline__ double butterfly64(double v) { return swap_add<32>(swap_add<16>(butterfly16(v))); }
// half-wave form: lanes 0-31 and 32-63 each reduce their own 32 values
__device__ __forceinline__ double butterfly32(double v) { return swap_add<16>(butterfly16(v)); }
// v of lane `lane` (compile-time constant) in every lane: v_readlane, no LDS-pipe trip
template <int LANE> __device__ __forceinline__ double lane_value(double v)
{
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, LANE), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), LANE);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
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
// The canonical dot ("dot order")
//   level 0   every aligned group of 64 consecutive elements: xor butterfly (above)
//             -> one partial per 64 elements, written by the producing kernel
//             (in SpMV: one per chunk, by the wave that owns the chunk);
//   level 1   every aligned group of 4 partials (256 elements): ((q0+q1)+q2)+q3;
//   level 2   1024 threads: thread t adds level-1 values t, t+1024, ... in order; each
//             wave butterflies; the 16 wave sums are added in wave order.
// Levels 1 and 2 run in ONE workgroup (reduce_final_1024); level 1 exists so that a
// thread's inputs are 32 contiguous bytes (two 16-B loads instead of four 8-B loads).
// Missing tail elements / partials count as +0.0: the partial array always holds
// 4*ceil(n/256) entries and producers zero the tail.
// =============================================================================
__device__ __forceinline__ double level1(const double* __restrict__ q, uint32_t i)
{
  const double2 a = *reinterpret_cast<const double2*>(q + 4u * (size_t)i);
  const double2 b = *reinterpret_cast<const double2*>(q + 4u * (size_t)i + 2u);
  return ((a.x + a.y) + b.x) + b.y;
}

// m = number of level-1 values = ceil(n/256); q holds 4*m level-0 partials -- or, l1 != 0, the m level-1 values
// themselves (a producer whose waves own whole 256-groups forms them in registers: cg_update_r_k; the single CU
// that runs this reduction then moves a quarter of the bytes through its one vector-memory pipe)
__device__ __forceinline__ double reduce_final_1024(uint32_t m, const double* __restrict__ q,
    double* lds16, int l1 = 0)
{
  double s   = 0.0;
  uint32_t i = threadIdx.x;
  if (l1) {
    for (; i + 7u * 1024u < m; i += 8u * 1024u) {
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = q[i + (uint32_t)u * 1024u];
#pragma unroll
      for (int u = 0; u < 8; u++) s = s + a[u];
    }
    for (; i < m; i += 1024u) s = s + q[i];
  } else {
  for (; i + 7u * 1024u < m; i += 8u * 1024u) { // 16 independent 16-B loads in flight
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = level1(q, i + (uint32_t)u * 1024u);
#pragma unroll
    for (int u = 0; u < 8; u++) s = s + a[u];
  }
  for (; i < m; i += 1024u) s = s + level1(q, i);
  }
  s = butterfly64(s);
  if ((threadIdx.x & 63u) == 0) lds16[threadIdx.x >> 6] = s;
  __syncthreads();
  double total = lds16[0];
#pragma unroll
  for (int w = 1; w < 16; w++) total = total + lds16[w];
  return total; // every thread returns the same value
}

// The scalar steps of solveCG.  MODE 0: r.r of the prologue (src/CGSolver.c:98-103) and the
// loop test for k = 1.  MODE 1: the loop test for the next k (:107), and if it passes
// the r.r / beta of that iteration (:111-113,:116).  MODE 2: p.Ap -> alpha (:124-126).
// `in`: the control block as it was when the kernel started (thread 0 fetches it next to the partial loads, so that
// the step does not begin with a second, dependent round trip to memory); only the changed fields are stored.
template <int MODE>
__device__ __forceinline__ void cg_apply(CgScalars* S, const CgScalars& in, double total, double* rr_hist,
    double* pAp_hist, int defer_x)
{
  if (MODE == 0) {
    const int sn = !(sqrt(total) > in.eps);
    S->rr        = total;
    S->stop_next = sn;
    if (in.n_rr < in.hist_cap) rr_hist[in.n_rr] = total;
    S->n_rr = in.n_rr + 1;
    if (1 < in.itermax && !sn) S->iters = 1;
    else S->stop = 1;
  } else if (MODE == 1) {
    if (in.iters + 1 < in.itermax && !in.stop_next) {
      const double old = in.rr;
      S->rr_old        = old;
      S->rr            = total;
      S->beta          = total / old;
      S->stop_next     = !(sqrt(total) > in.eps);
      S->iters         = in.iters + 1;
      if (in.n_rr < in.hist_cap) rr_hist[in.n_rr] = total;
      S->n_rr = in.n_rr + 1;
    } else {
      S->stop = 1;
    }
    if (defer_x) S->x_pending = 1; // the body that just ran left "x += alpha p" to the next p update
  } else {
    S->x_pending    = 0; // consumed by the p update that preceded this SpMV
    S->snap_rr = in.rr, S->snap_iters = in.iters, S->snap_stop_next = in.stop_next;
    S->pAp          = total;
    const double al = in.rr / total;
    S->alpha        = al;
    S->neg_alpha    = -al;
    if (in.n_pAp < in.hist_cap) pAp_hist[in.n_pAp] = total;
    S->n_pAp = in.n_pAp + 1;
  }
}
template <int MODE>
__device__ __forceinline__ void cg_apply(CgScalars* S, double total, double* rr_hist, double* pAp_hist, int defer_x)
{
  const CgScalars in = *S;
  cg_apply<MODE>(S, in, total, rr_hist, pAp_hist, defer_x);
}
// the control block into registers, pinned in front of whatever follows
__device__ __forceinline__ CgScalars cg_fetch(const CgScalars* S)
{
  const CgScalars in = *S;
  asm volatile("" ::"v"(in.rr), "v"(in.eps), "v"(in.stop), "v"(in.stop_next), "v"(in.iters), "v"(in.n_rr), "v"(in.n_pAp),
      "v"(in.itermax), "v"(in.hist_cap));
  return in;
}

// SpMV epilogue: y store and, when DOT, the chunk's level-0 partial of p.Ap (a chunk IS a
// 64-group of the output vector).  No barrier: each wave finishes on its own.
template <bool DOT>
__device__ __forceinline__ void spmv_epilogue(uint32_t chunk, uint32_t lane, double acc,
    const double* __restrict__ x, double* __restrict__ y, uint32_t nr, double* __restrict__ dotPartials)
{
  const uint32_t row = chunk * 64u + lane;
  if (row < nr) y[row] = acc;
  if (DOT) {
    double t = row < nr ? x[row] * acc : 0.0;
    t        = butterfly64(t);
    if (lane == 0) dotPartials[chunk] = t;
  }
}

// =============================================================================
// Sell-C-sigma SpMV, C = 64, reference layout: one wavefront per chunk, lane k = row
// k of the chunk (reference loop: src/matrix-SCS.c:208-227, its inner k loop is our
// lane axis).  val/colInd are column-major inside the chunk, so each wave-instruction
// reads 512 B of val and 256 B of colInd, fully coalesced.  Each lane accumulates its
// row left to right exactly like the CPU loop.
// =============================================================================
constexpr uint32_t SCS_SLACK = 16 * 64; // zeroed elements behind val / colInd

template <int UNROLL, bool DOT, bool NT>
__global__ __launch_bounds__(256) void spmv_scs64(const uint32_t* __restrict__ chunkPtr,
    const uint32_t* __restrict__ chunkLens, const uint32_t* __restrict__ colInd,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
    uint32_t nr, uint32_t nChunks, uint32_t blocksPerXcd, double* __restrict__ dotPartials,
    const int* __restrict__ stop)
{
  const int stopped      = stop ? *stop : 0; // one wait covers this and the loads below
  const uint32_t nBlocks = (nChunks + 3u) >> 2;
  const uint32_t lb      = blocksPerXcd ? xcd_block(blockIdx.x, blocksPerXcd) : blockIdx.x;
  if (lb >= nBlocks || stopped) return; // uniform per workgroup
  const uint32_t chunk = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
  const uint32_t lane  = threadIdx.x & 63u;
  const bool active    = chunk < nChunks; // wave-uniform; with DOT an idle wave of the last block still joins the combine below
  if (!DOT && !active) return;
  double acc = 0.0;
  if (active) {
    const uint32_t cp  = chunkPtr[chunk];
    const uint32_t len = chunkLens[chunk];
    const double* v    = val + cp + lane;
    const uint32_t* c  = colInd + cp + lane;
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
  }
  // y, and with DOT the LEVEL-1 value of the block's four chunks (a block IS an aligned 256-group of the output vector):
  // the four waves' level-0 partials meet in LDS and are added ((q0 + q1) + q2) + q3 -- the scalar step then reads
  // n/256 doubles instead of n/64 through its one CU (which was 1.9 of its 4.3 us: profiles/r03_scalar_anatomy.txt)
  const uint32_t row = chunk * 64u + lane;
  if (active && row < nr) y[row] = acc;
  if (DOT) {
    __shared__ double sq[4];
    double t = (active && row < nr) ? x[row] * acc : 0.0;
    t        = butterfly64(t);
    if (lane == 0) sq[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) dotPartials[lb] = ((sq[0] + sq[1]) + sq[2]) + sq[3];
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
// CRS SpMV (reference loop: src/matrix-CRS.c:54-64), streaming the reference's own arrays
// (12 B per nonzero).  Row blocks are cut on the host so that a block's nonzeros fit one LDS
// tile (<= 2048 nonzeros, <= 256 rows): the workgroup streams val / col with coalesced
// non-temporal loads (all of a thread's loads issued before the first use), gathers x, and
// puts the PRODUCTS into LDS; thread r then adds row r's products left to right -- the CPU's
// order, hence the CPU's bits -- four LDS reads in flight at a time.  A row longer than the
// tile is walked tile by tile by thread 0 (still in order).  8 workgroups per CU.
// Measured against it on the same box (round 2, irregular stand-in 94 M nonzeros / HPCG 128^3 inside CG):
// a software-pipelined kernel with row sums carried from tile to tile, 64-row groups per wave and the
// p.Ap partials fused -- 260 vs 238 us and 167 vs 151 us, and fewer CG iterations/s even though it saves the
// separate dot pass (registers and LDS for the pipeline cost half the resident workgroups); and the
// unbatched summation loop of round 1 -- 159 vs 151 us.  DESIGN.md section 4.
// =============================================================================
constexpr int CRS_THREADS = 256;
constexpr int CRS_TILE  = 2048; // nonzeros per LDS tile (16 KiB: 8 workgroups per CU)
constexpr int CRS_BATCH = CRS_TILE / CRS_THREADS;

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
      uint32_t k = a;
      for (; k + 4u <= b; k += 4u) { // four LDS reads in flight, then the four adds in order
        const double d0 = prod[k], d1 = prod[k + 1u], d2 = prod[k + 2u], d3 = prod[k + 3u];
        sum = (((sum + d0) + d1) + d2) + d3;
      }
      for (; k < b; k++) sum = sum + prod[k];
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

// The same product with the nonzeros cut into EQUAL windows (the default where no row is longer than CRS_SPLIT_MAXROW):
// tile lb owns the rows that START in nonzeros [lb T, (lb + 1) T) and streams the fixed window [lb T, lb T + 2048), which
// covers them to the end of the last one (T <= 2049 - longest row, a multiple of 64; the first few elements may belong to
// the previous tile's last row: multiplied, never summed).  What this buys is the dependence chain: the window is known
// from blockIdx alone, so the 16 stream loads of a thread leave at kernel entry together with the tile's row range
// (tileRow: first row starting at or behind lb T, cut on the host), the gathers and the thread's row extent follow as
// the second round trip, then LDS and the in-order sums -- two dependent round trips where spmv_crs_stream has four
// (row block -> rowPtr of its ends -> stream -> gather), all tiles the same size.  Row sums are the same left-to-right
// adds of the same products: same bits.  A tile may hold more than 256 rows (short or empty rows): threads loop.
// Measured (one box, stand-alone / inside CG): HPCG 128^3 130.0 / 141.9 us against the row-block kernel's 139.5 / 152 us;
// irregular stand-in 201.8 against 213.1 us (tools/crs_ab.py).  Tiles of 3072 / 4096 nonzeros (6 / 5 workgroups per CU):
// 127.3 / 141.2 us against 124.7 us with 2048 on the same box -- kept at 2048.
constexpr uint32_t CRS_SPLIT_MAXROW = 1025;
__global__ __launch_bounds__(CRS_THREADS) void spmv_crs_split(
    const uint32_t* __restrict__ tileRow, const uint32_t* __restrict__ rowPtr,
    const uint32_t* __restrict__ colInd, const double* __restrict__ val,
    const double* __restrict__ x, double* __restrict__ y, uint32_t nTiles, uint32_t T, uint32_t nnz,
    uint32_t blocksPerXcd, const int* __restrict__ stop)
{
  __shared__ double prod[CRS_TILE];
  const uint32_t lb = xcd_block(blockIdx.x, blocksPerXcd);
  if (lb >= nTiles) return;
  const uint32_t base = lb * T, t = threadIdx.x;
  // a thread takes PAIRS of neighbouring nonzeros (16 B of val, 8 B of colInd per load: half the vector-memory instructions
  // of an element per load; base is a multiple of 64 elements, the arrays end in 64 zeroed elements of slack)
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
  constexpr int PAIRS = CRS_BATCH / 2;
  dbl2 v[PAIRS], xv[PAIRS];
  uint2v c[PAIRS];
#pragma unroll
  for (int u = 0; u < PAIRS; u++) {
    const uint32_t k = base + 2u * t + (uint32_t)u * 2u * CRS_THREADS;
    v[u] = dbl2{ 0.0, 0.0 }, c[u] = uint2v{ 0u, 0u };
    if (k < nnz) v[u] = stream_load(reinterpret_cast<const dbl2*>(val + k)), c[u] = stream_load(reinterpret_cast<const uint2v*>(colInd + k));
  }
  const int stopped = stop ? *stop : 0;
  const uint32_t r0 = tileRow[lb], r1 = tileRow[lb + 1];
  if (stopped) return;
#pragma unroll
  for (int u = 0; u < PAIRS; u++) xv[u] = dbl2{ x[c[u].x], x[c[u].y] };
  uint32_t r = r0 + t, a = 0, b = 0;
  if (r < r1) a = rowPtr[r] - base, b = rowPtr[r + 1] - base;
#pragma unroll
  for (int u = 0; u < PAIRS; u++)
    *reinterpret_cast<dbl2*>(&prod[2u * t + (uint32_t)u * 2u * CRS_THREADS]) = dbl2{ v[u].x * xv[u].x, v[u].y * xv[u].y };
  __syncthreads();
  while (r < r1) {
    double sum = 0.0;
    uint32_t k = a;
    for (; k + 4u <= b; k += 4u) { // four LDS reads in flight, then the four adds in order
      const double d0 = prod[k], d1 = prod[k + 1u], d2 = prod[k + 2u], d3 = prod[k + 3u];
      sum = (((sum + d0) + d1) + d2) + d3;
    }
    for (; k < b; k++) sum = sum + prod[k];
    y[r] = sum;
    r += CRS_THREADS;
    if (r < r1) a = rowPtr[r] - base, b = rowPtr[r + 1] - base;
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
// If x != NULL and the previous body left its "x = x + alpha p" (:127) pending, it is
// applied here, where the old p is in registers anyway (saves one read of p per
// iteration); the arithmetic and its order per element are unchanged.
//
// BETA != 0 (which = 0 only; 1024-thread workgroups): the beta step / loop test the previous body left owing (:107, :111-113,
// :116) rides at the head of this launch, taken by EVERY workgroup -- nobody waits for anybody; workgroup 0 also records it
// in the control block (cg_apply<1>).  BETA = 1 (one rank): each workgroup reduces the m level-1 values of r.r itself in
// the canonical order; BETA = 2 (several ranks on the communicator's collectives): r.r is already reduced and all-reduced
// into S->local by the two launches before.  One launch per dot fewer either way; same operations, same order, same bits.
template <int BETA>
__global__ __launch_bounds__(1024) void cg_update_p(uint32_t n, const double* __restrict__ r,
    double* p, double* x, CgScalars* S, int which, uint32_t m, const double* __restrict__ rrL1, double* __restrict__ rr_hist)
{
  const uint32_t n2     = n >> 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const double2* r2     = reinterpret_cast<const double2*>(r);
  double2* p2           = reinterpret_cast<double2*>(p);
  double2* x2           = reinterpret_cast<double2*>(x);
  uint32_t i            = blockIdx.x * blockDim.x + threadIdx.x;
  const bool useX       = x != nullptr && which == 0;
  // Two elements (i, i + stride) per step, and the first pair goes in flight together with
  // the control block (stop flag, beta, alpha) instead of behind it: clamped indices, so the
  // loads are unconditional.
  const uint32_t last = n2 ? n2 - 1u : 0u;
  double2 a0 = { 0.0, 0.0 }, b0 = a0, x0 = a0, a1 = a0, b1 = a0, x1 = a0;
  auto load = [&](uint32_t j, double2& a, double2& b, double2& xv) {
    a = r2[j];
    b = which == 0 ? p2[j] : a;
    if (useX) xv = x2[j];
  };
  if (n2) load(min(i, last), a0, b0, x0), load(min(i + stride, last), a1, b1, x1);
  double beta, alpha;
  bool owed;
  if (BETA) {
    __shared__ double lds16[16];
    const bool recorder = blockIdx.x == 0 && threadIdx.x == 0;
    CgScalars in;
    if (recorder) in = *S; // (only the recorder needs all of it)
    // (stop: a late workgroup may already see the flag the recorder raises when the loop test fails -- the same outcome)
    const int stopped = S->stop, iters = S->snap_iters, itermax = S->itermax, stop_next = S->snap_stop_next;
    const double rr = S->snap_rr, total2 = S->local;
    alpha = S->alpha;
    const double total = BETA == 1 ? reduce_final_1024(m, rrL1, lds16, 1) : total2;
    if (stopped) return;
    if (recorder && !in.stop) cg_apply<1>(S, in, total, rr_hist, (double*)nullptr, 1);
    if (!(iters + 1 < itermax && !stop_next)) return; // the loop test failed (cg_apply<1> has raised the stop flag): p stays
    beta = total / rr; // (= cg_apply<1>'s S->beta)
    owed = useX;       // (the body that left the step owing also left its x update: defer_x)
  } else {
    const int stopped = S->stop;
    beta  = which == 0 ? S->beta : 0.0;
    owed  = useX && S->x_pending;
    alpha = S->alpha;
    if (stopped) return;
  }
  auto finish = [&](uint32_t j, const double2& a, const double2& b, double2 xv) {
    if (owed) {
      xv.x = xv.x + alpha * b.x;
      xv.y = xv.y + alpha * b.y;
      x2[j] = xv;
    }
    double2 o;
    o.x = a.x + beta * b.x;
    o.y = a.y + beta * b.y;
    p2[j] = o;
  };
  for (; i < n2; i += 2u * stride) {
    const bool second = i + stride < n2;
    finish(i, a0, b0, x0);
    if (second) finish(i + stride, a1, b1, x1);
    const uint32_t nx = i + 2u * stride;
    if (nx < n2) load(nx, a0, b0, x0), load(min(nx + stride, last), a1, b1, x1);
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double bb = which == 0 ? p[n - 1] : r[n - 1];
    if (owed) x[n - 1] = x[n - 1] + alpha * bb;
    p[n - 1] = r[n - 1] + beta * bb;
  }
}

// the owed "x = x + alpha p" of the LAST body that ran (nobody comes after it)
// (p0 / p1: with the p update inside the SpMV, body k leaves p_k in buffer k & 1 and n_pAp bodies have run; otherwise p0 == p1)
__global__ __launch_bounds__(256) void cg_x_finalize(uint32_t n, double* x, const double* __restrict__ p0,
    const double* __restrict__ p1, const CgScalars* __restrict__ S)
{
  if (!S->x_pending) return;
  const double* __restrict__ p = (S->n_pAp & 1) ? p1 : p0;
  const double alpha    = S->alpha;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = x[i] + alpha * p[i];
}

// ---- dot-producing vector kernels -------------------------------------------------------
// A wave covers a "span" of 128 consecutive elements per step with 16-B loads: lane l
// holds elements 2l, 2l+1 (its in-lane add is butterfly offset 1, lane-xor 1..16 are
// offsets 2..32), so lanes 0-31 / 32-63 produce the level-0 partials 2s / 2s+1.  Spans
// beyond n write +0.0 so the partial array is complete up to 4*ceil(n/256).
// OP 0: dot(a, b)                                       (ddot, src/solver.c:41-62)
// OP 1: x += alpha p ; r -= alpha Ap ; dot(r, r)        (src/CGSolver.c:127-128 + :112)
// OP 2: r = b - Ap ; dot(r, r)                          (src/CGSolver.c:97-98)
// (the fused loop's r update is cg_update_r_k below)
template <int OP>
__global__ __launch_bounds__(256) void dot_spans_k(uint32_t n, const double* a, const double* b,
    double* x, double* r, const CgScalars* __restrict__ S, double* __restrict__ partials,
    const int* __restrict__ stop)
{
  const uint32_t lane   = threadIdx.x & 63u;
  const uint32_t nSpans = ((n + 255u) >> 8) * 2u; // whole 256-groups
  const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
  if (stop && *stop) return;
  double alpha = 0.0, nalpha = 0.0;
  if (OP == 1) alpha = S->alpha, nalpha = -alpha;
  for (uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nSpans; s += nWaves) {
    const uint32_t e = s * 128u + lane * 2u;
    double t         = 0.0;
    if (OP == 0) {
      if (e + 1 < n) {
        const double2 av = *reinterpret_cast<const double2*>(a + e);
        const double2 bv = *reinterpret_cast<const double2*>(b + e);
        t                = av.x * bv.x + av.y * bv.y;
      } else if (e < n) {
        t = a[e] * b[e] + 0.0;
      }
    } else if (OP == 1) { // a = p, b = Ap
      if (e + 1 < n) {
        double2 xv       = *reinterpret_cast<double2*>(x + e);
        const double2 pv = *reinterpret_cast<const double2*>(a + e);
        double2 rv       = *reinterpret_cast<double2*>(r + e);
        const double2 av = *reinterpret_cast<const double2*>(b + e);
        xv.x = xv.x + alpha * pv.x;
        xv.y = xv.y + alpha * pv.y;
        rv.x = rv.x + nalpha * av.x;
        rv.y = rv.y + nalpha * av.y;
        *reinterpret_cast<double2*>(x + e) = xv;
        *reinterpret_cast<double2*>(r + e) = rv;
        t = rv.x * rv.x + rv.y * rv.y;
      } else if (e < n) {
        x[e]            = x[e] + alpha * a[e];
        const double rn = r[e] + nalpha * b[e];
        r[e]            = rn;
        t               = rn * rn + 0.0;
      }
    } else { // a = b (rhs), b = Ap
      if (e + 1 < n) {
        const double2 bv = *reinterpret_cast<const double2*>(a + e);
        const double2 av = *reinterpret_cast<const double2*>(b + e);
        double2 rv;
        rv.x = bv.x + -1.0 * av.x;
        rv.y = bv.y + -1.0 * av.y;
        *reinterpret_cast<double2*>(r + e) = rv;
        t = rv.x * rv.x + rv.y * rv.y;
      } else if (e < n) {
        const double rn = a[e] + -1.0 * b[e];
        r[e]            = rn;
        t               = rn * rn + 0.0;
      }
    }
    t = butterfly32(t);
    if ((lane & 31u) == 0) partials[s * 2u + (lane >> 5)] = t;
  }
}

// dot(a, b) as LEVEL-1 values (one per 256 elements): the dot pass of the fused loop behind an SpMV kernel that cannot emit
// p.Ap itself (native CRS, generic-C Sell-C-sigma).  A wave owns whole 256-groups, as in cg_update_r_k below: same additions in the
// same order as dot_spans_k<0> followed by level1(), a quarter of the values for the scalar step -- and level-1 values are what
// lets that step ride in the r update (cg_update_r_k<1>).
__global__ __launch_bounds__(1024) void dot_l1_k(uint32_t n, const double* __restrict__ a, const double* __restrict__ b,
    double* __restrict__ l1out, const int* __restrict__ stop)
{
  const uint32_t lane    = threadIdx.x & 63u;
  const uint32_t nGroups = (n + 255u) >> 8;
  const uint32_t nWaves  = gridDim.x * (blockDim.x >> 6);
  if (stop && *stop) return;
  auto combine = [&](double t0, double t1) { // halves of t0: q0, q1; of t1: q2, q3
    const double q0 = lane_value<0>(t0), q1 = lane_value<32>(t0), q2 = lane_value<0>(t1), q3 = lane_value<32>(t1);
    return ((q0 + q1) + q2) + q3;
  };
  for (uint32_t gI = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); gI < nGroups; gI += nWaves) {
    double t[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t e = gI * 256u + (uint32_t)h * 128u + lane * 2u;
      double tt        = 0.0;
      if (e + 1 < n) {
        const double2 av = *reinterpret_cast<const double2*>(a + e);
        const double2 bv = *reinterpret_cast<const double2*>(b + e);
        tt               = av.x * bv.x + av.y * bv.y;
      } else if (e < n) {
        tt = a[e] * b[e] + 0.0;
      }
      t[h] = butterfly32(tt);
    }
    const double v = combine(t[0], t[1]);
    if (lane == 0) l1out[gI] = v;
  }
}

// r -= alpha Ap and r.r of the fused loop (src/CGSolver.c:128 + :112), once per CG iteration.  A wave owns whole
// 256-groups (two adjacent spans), so it forms the group's LEVEL-1 value ((q0 + q1) + q2) + q3 in registers and the
// scalar step that follows reads n/256 doubles instead of n/64 (reduce_final_1024, l1).  The first group's loads go
// in flight together with the stop flag / alpha instead of behind them.  Same arithmetic, same order, same bits as
// a per-span r update followed by level1().
//
// ALPHA (one rank, 1024-thread workgroups, the p.Ap producer emits level-1 values): the alpha step rides in this launch --
// EVERY workgroup reduces the m level-1 values of p.Ap itself (reduce_final_1024: the canonical order, so all of them
// hold the same bits), divides, and goes on; workgroup 0 also records the step in the control block (cg_apply<2>).  Nobody
// waits for anybody -- unlike the lead kernels of `fused = 3`, where workgroup 0 published and the others polled --, the
// price is m * 8 bytes of L2 reads per workgroup (64 KB at 128^3, 512 workgroups) behind the first group's r / Ap loads,
// which are already in flight.  One launch and its boundary fewer per loop body; same operations, same order, same bits.
// ALPHA = 2 (several ranks on the communicator's collectives): p.Ap is already reduced and all-reduced into S->local by the
// two launches before; every workgroup divides, workgroup 0 records -- the third launch of that dot goes.
template <int ALPHA>
__global__ __launch_bounds__(1024) void cg_update_r_k(uint32_t n, const double* __restrict__ Ap, double* r,
    CgScalars* S, double* __restrict__ l1out, const int* stop, uint32_t m, const double* __restrict__ pApL1,
    double* __restrict__ rr_hist, double* __restrict__ pAp_hist)
{
  const uint32_t lane    = threadIdx.x & 63u;
  const uint32_t nGroups = (n + 255u) >> 8;
  const uint32_t nWaves  = gridDim.x * (blockDim.x >> 6);
  uint32_t gI            = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  auto full = [&](uint32_t gg) { return gg < nGroups && gg * 256u + 256u <= n; }; // wave-uniform
  double2 r0 = { 0.0, 0.0 }, a0 = r0, r1 = r0, a1 = r0;
  bool have  = full(gI);
  if (have) {
    const uint32_t e0 = gI * 256u + lane * 2u, e1 = e0 + 128u;
    r0 = *reinterpret_cast<const double2*>(r + e0), a0 = *reinterpret_cast<const double2*>(Ap + e0);
    r1 = *reinterpret_cast<const double2*>(r + e1), a1 = *reinterpret_cast<const double2*>(Ap + e1);
  }
  double nalpha;
  if (ALPHA) {
    __shared__ double lds16[16];
    const bool recorder = blockIdx.x == 0 && threadIdx.x == 0;
    CgScalars in;
    if (recorder) in = *S; // (prefetched next to the partials; only the recorder needs all of it)
    const int stopped  = S->stop;
    const double rr    = S->rr;
    const double total2 = S->local;
    const double total  = ALPHA == 1 ? reduce_final_1024(m, pApL1, lds16, 1) : total2;
    if (stopped) return;
    nalpha = -(rr / total); // (= cg_apply<2>'s neg_alpha: -(in.rr / total))
    if (recorder) cg_apply<2>(S, in, total, rr_hist, pAp_hist, 0);
  } else {
    if (stop && *stop) return;
    nalpha = -S->alpha;
  }
  auto combine = [&](double t0, double t1) { // halves of t0: q0, q1; of t1: q2, q3
    const double q0 = lane_value<0>(t0), q1 = lane_value<32>(t0), q2 = lane_value<0>(t1), q3 = lane_value<32>(t1);
    return ((q0 + q1) + q2) + q3;
  };
  while (have) {
    const uint32_t e0 = gI * 256u + lane * 2u, e1 = e0 + 128u;
    r0.x = r0.x + nalpha * a0.x, r0.y = r0.y + nalpha * a0.y;
    r1.x = r1.x + nalpha * a1.x, r1.y = r1.y + nalpha * a1.y;
    *reinterpret_cast<double2*>(r + e0) = r0;
    *reinterpret_cast<double2*>(r + e1) = r1;
    const double v = combine(butterfly32(r0.x * r0.x + r0.y * r0.y), butterfly32(r1.x * r1.x + r1.y * r1.y));
    if (lane == 0) l1out[gI] = v;
    gI += nWaves;
    have = full(gI);
    if (have) {
      const uint32_t f0 = gI * 256u + lane * 2u, f1 = f0 + 128u;
      r0 = *reinterpret_cast<const double2*>(r + f0), a0 = *reinterpret_cast<const double2*>(Ap + f0);
      r1 = *reinterpret_cast<const double2*>(r + f1), a1 = *reinterpret_cast<const double2*>(Ap + f1);
    }
  }
  for (; gI < nGroups; gI += nWaves) { // the last, partial group
    double t[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t e = gI * 256u + (uint32_t)h * 128u + lane * 2u;
      double tt        = 0.0;
      if (e + 1 < n) {
        double2 rv       = *reinterpret_cast<double2*>(r + e);
        const double2 av = *reinterpret_cast<const double2*>(Ap + e);
        rv.x = rv.x + nalpha * av.x;
        rv.y = rv.y + nalpha * av.y;
        *reinterpret_cast<double2*>(r + e) = rv;
        tt = rv.x * rv.x + rv.y * rv.y;
      } else if (e < n) {
        const double rn = r[e] + nalpha * Ap[e];
        r[e]            = rn;
        tt              = rn * rn + 0.0;
      }
      t[h] = butterfly32(tt);
    }
    const double v = combine(t[0], t[1]);
    if (lane == 0) l1out[gI] = v;
  }
}

// stand-alone level 2 (sb_reduce_final / unfused path)
__global__ __launch_bounds__(1024) void reduce_final_k(uint32_t m, const double* __restrict__ q,
    double* __restrict__ out, const int* __restrict__ stop)
{
  __shared__ double lds16[16];
  if (stop && *stop) return;
  const double total = reduce_final_1024(m, q, lds16);
  if (threadIdx.x == 0) *out = total;
}

// =============================================================================
// In-kernel all-reduce of ONE double over peer-mapped memory (several ranks).
// A dot product on P ranks is: local reduce | all-reduce of 8 bytes | scalar step.  Through
// RCCL that is three dependent launches (~25 us); here the scalar step's own workgroup does
// the exchange: every rank owns a small FINE-GRAINED buffer that all peers have mapped (HIP
// IPC, xGMI); thread r stores this rank's value and then the sequence number into peer r's
// buffer (system-scope release), thread r then waits for rank r's pair in the own buffer
// (system-scope acquire), and the P values are added pairwise in rank order -- the same tree
// on every rank, so all ranks continue with identical bits.  Two slot sets, used alternately:
// a rank cannot be two exchanges ahead of a peer, because each exchange waits for all.
// The wait is bounded (wall clock); a timeout raises CgScalars::p2p_error and stops the loop.
// =============================================================================
constexpr int P2P_MAX = 16; // ranks per communicator this path supports (one node)
struct P2PSlot {
  unsigned long long bits; // the double
  unsigned long long seq;
};
struct P2PView {
  int rank, size;
  long long timeoutTicks; // bound of every wait, in wall_clock64 ticks (100 MHz); SB_P2P_TIMEOUT_MS
  P2PSlot* peer[P2P_MAX]; // peer[r]: rank r's buffer, P2PSlot[2][P2P_MAX], as mapped in this process
};
// A wait is bounded so that a dead peer ends the run with a message instead of hanging the GPU; the bound
// is generous (default 30 s inside CG, SB_P2P_TIMEOUT_MS) because a merely LATE peer (first-kernel load, OS
// jitter) must not be fatal -- RCCL would simply wait.  The set-up self-tests use 5 s.
constexpr long long P2P_TICKS_PER_MS = 100000ll;
// A rank whose bounded wait ran out (or that was told so by a peer) leaves the loop -- and must not leave the others sitting
// out their own 30 s: from then on it writes P2P_POISON where it would have published a sequence number (all-reduce slots,
// halo flags), and every wait treats that value as "the peer has failed": the whole job ends within microseconds of the
// first failure, and only the rank that saw the cause reports it as one (CgScalars::p2p_error for the all-reduce, the halo
// plan's err for the halo waits: 1 = the wait ran out HERE, 2 = another rank reported a failure).
constexpr unsigned long long P2P_POISON = ~0ull;
// bounded wait for *f == seq: 0 it arrived, 1 timed out, 2 the peer has failed
__device__ __forceinline__ int p2p_wait(const unsigned long long* f, unsigned long long seq, long long timeoutTicks)
{
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned long long got = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (got == seq) return 0;
    if (got == P2P_POISON) return 2;
    if (wall_clock64() - t0 > timeoutTicks) return 1;
    __builtin_amdgcn_s_sleep(4);
  }
}
// a halo wait failed (how: p2p_wait's 1 / 2): raise the halo plan's error flag (the cause; the rank's next push / scalar
// kernels see it and poison what they would have published) and the stop flag
__device__ __forceinline__ void halo_wait_failed(int how, int* err, int* stopw)
{
  atomicExch(err, how);
  if (stopw) atomicExch(stopw, 1); // the loop must not go on iterating on a stale halo
}

// every thread of the workgroup calls this with the same `mine`; returns the same sum in every thread
__device__ __forceinline__ double p2p_allreduce_sum(const P2PView* pv, double mine, unsigned long long seq,
    double* sh /* >= P2P_MAX doubles of LDS */, int* err)
{
  const int t = (int)threadIdx.x, P = pv->size;
  const unsigned par = (unsigned)(seq & 1ull);
  if (t < P) {
    P2PSlot* dst = pv->peer[t] + par * P2P_MAX + pv->rank;
    __hip_atomic_store(&dst->bits, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    P2PSlot* src       = pv->peer[pv->rank] + par * P2P_MAX + t;
    const int how = p2p_wait(&src->seq, seq, pv->timeoutTicks);
    sh[t] = !how ? __longlong_as_double((long long)__hip_atomic_load(&src->bits, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM))
                 : 0.0;
    if (how) atomicCAS(err, 0, how); // 1: timed out, 2: the peer has failed (the first cause stays)
  }
  __syncthreads();
  // pairwise tree in rank order: ((v0+v1)+(v2+v3))+... (an odd tail moves up unchanged)
  double v[P2P_MAX];
#pragma unroll
  for (int i = 0; i < P2P_MAX; i++) v[i] = i < P ? sh[i] : 0.0;
  int n = P;
  while (n > 1) {
    const int h = n >> 1;
#pragma unroll
    for (int i = 0; i < P2P_MAX / 2; i++)
      if (i < h) v[i] = v[2 * i] + v[2 * i + 1];
    if (n & 1) v[h] = v[n - 1];
    n = h + (n & 1);
  }
  __syncthreads();
  return v[0];
}

__global__ __launch_bounds__(64) void p2p_selftest_k(const P2PView* pv, unsigned long long seq, double mine,
    double* out, int* err)
{
  __shared__ double sh[P2P_MAX];
  const double r = p2p_allreduce_sum(pv, mine, seq, sh, err);
  if (threadIdx.x == 0) *out = r;
}

// =============================================================================
// Halo exchange over peer-mapped memory (same ingredients as p2p_allreduce_sum).  Every rank
// owns a fine-grained staging area [2][externalCount] + flags [2][P2P_MAX] that its neighbours
// have mapped (HIP IPC).  halo_push_k packs x[elementsToSend] (src/comm.c:635-638) and stores
// every value straight into the slot the receiver expects it in (its rdispl + offset); the
// last workgroup to finish raises the sequence flag at every destination (system-scope
// release).  halo_pull_k (one workgroup per source) waits for its source's flag (bounded) and
// copies the block into the tail of x (src/comm.c:640-648 receives there).  Two launches, no
// RCCL call in the loop.  Parity-alternating areas: between two exchanges lie two all-reduces,
// so no rank is more than one exchange ahead of a neighbour.
// =============================================================================
struct HaloPush {
  uint32_t n;               // elements to send
  int ndest, rank;
  const uint32_t* packIdx;  // row (in the vector's order) of each element
  const uint32_t* slot;     // its position in the receiver's staging area
  const uint8_t* dest;      // which destination (index into the arrays below)
  unsigned int* done;       // workgroups finished (last-block protocol)
  unsigned long long* stage[P2P_MAX]; // destination i's staging area, as mapped here
  unsigned long long* flag[P2P_MAX];  // destination i's flags
  uint32_t ext[P2P_MAX];              // destination i's externalCount (area stride)
  long long timeoutTicks;             // bound of the receivers' waits (halo_pull_k, HALO SpMV)
  unsigned long long dropSeq;         // test hook (SB_TEST_DROP_PUSH_RANK / _AT): the exchange this rank does not announce
  const int* err;                     // the halo plan's error flag (a failed wait of THIS rank) ...
  const int* p2pErr;                  // ... and the control block's (a failed all-reduce; NULL outside CG): push kernels poison on either
};
__device__ __forceinline__ bool halo_rank_failed(const HaloPush& hp)
{
  return (hp.err && *hp.err) || (hp.p2pErr && *hp.p2pErr);
}

// the push of one workgroup out of nBlocks (its own kernel below; or the first workgroups of the HALO SpMV)
// FUSEP: the values to send are not in memory yet -- the SpMV that follows forms p_new = r + beta p_old while it stages its
// windows (pack.hip.h: spmv_prog_fusep) -- so the push forms them itself, with the same expression: x = p_old here.
// this rank has failed: every peer that waits for its all-reduce contribution (either parity) leaves at once
__device__ __forceinline__ void p2p_poison_allreduce(const P2PView* pv)
{
  const int t = (int)threadIdx.x;
  if (t < pv->size)
    for (int par = 0; par < 2; par++)
      __hip_atomic_store(&(pv->peer[t] + par * P2P_MAX + pv->rank)->seq, P2P_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <bool FUSEP = false>
__device__ __forceinline__ void halo_push_block(const HaloPush& hp, const double* __restrict__ x, unsigned long long seq,
    uint32_t block, uint32_t nBlocks, const double* __restrict__ r = nullptr, double beta = 0.0)
{
  const unsigned par    = (unsigned)(seq & 1ull);
  const uint32_t stride = nBlocks * blockDim.x;
  for (uint32_t i = block * blockDim.x + threadIdx.x; i < hp.n; i += stride) {
    const uint32_t d = hp.dest[i];
    const uint32_t j = hp.packIdx[i];
    const double v   = FUSEP ? r[j] + beta * x[j] : x[j];
    __hip_atomic_store(hp.stage[d] + (size_t)par * hp.ext[d] + hp.slot[i],
        (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system(); // this thread's stores are out before its workgroup counts itself done
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(hp.done, 1u);
    if (prev == nBlocks - 1u) { // every workgroup has pushed: tell the receivers
      *hp.done = 0u;
      __threadfence_system();
      if (seq != hp.dropSeq) // (test hook: this rank "forgets" to announce exchange dropSeq; 0 = never)
        for (int d = 0; d < hp.ndest; d++)
          __hip_atomic_store(hp.flag[d] + par * P2P_MAX + hp.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// this rank has failed: the neighbours that wait for its halo block (either parity) leave at once.  One thread.
__device__ __forceinline__ void halo_poison_flags(const HaloPush& hp)
{
  for (int d = 0; d < hp.ndest; d++)
    for (unsigned par = 0; par < 2u; par++)
      __hip_atomic_store(hp.flag[d] + par * P2P_MAX + hp.rank, P2P_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void halo_push_k(HaloPush hp, const double* __restrict__ x,
    unsigned long long seq, const int* __restrict__ stop)
{
  if (stop && *stop) { // the same decision on every rank (the loop test is all-reduced) -- unless this rank has failed
    if (blockIdx.x == 0 && threadIdx.x == 0 && halo_rank_failed(hp)) halo_poison_flags(hp);
    return;
  }
  halo_push_block(hp, x, seq, blockIdx.x, gridDim.x);
}
// the push in front of spmv_prog_fusep: p_new = r + beta p_old of the boundary rows, formed here (which: k = 1, beta = 0, x = r)
__global__ __launch_bounds__(256) void halo_push_fusep_k(HaloPush hp, const double* __restrict__ pold, const double* __restrict__ r,
    const CgScalars* __restrict__ S, int which, unsigned long long seq)
{
  if (S->stop) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && halo_rank_failed(hp)) halo_poison_flags(hp);
    return;
  }
  halo_push_block<true>(hp, pold, seq, blockIdx.x, gridDim.x, r, which ? 0.0 : S->beta);
}

__global__ __launch_bounds__(256) void halo_pull_k(const int* __restrict__ srcRank, const int* __restrict__ rdispl,
    const int* __restrict__ rcount, const unsigned long long* stage, const unsigned long long* flags,
    uint32_t ext, double* __restrict__ xTail, unsigned long long seq, int* err, int* stop, long long timeoutTicks)
{
  __shared__ int ok;
  if (stop && *stop) return;
  const unsigned par = (unsigned)(seq & 1ull);
  const int j        = (int)blockIdx.x;
  if (threadIdx.x == 0) {
    const unsigned long long* f = flags + par * P2P_MAX + srcRank[j];
    const int how = p2p_wait(f, seq, timeoutTicks);
    if (how) halo_wait_failed(how, err, stop);
    ok = !how;
  }
  __syncthreads();
  if (!ok) return;
  // plain loads: thread 0's system-scope acquire + the barrier order them after the sender's
  // stores (one atomic load per element would serialise into one ~2 us round trip each)
  const double* src = reinterpret_cast<const double*>(stage + (size_t)par * ext + rdispl[j]);
  double* dst       = xTail + rdispl[j];
  for (int i = (int)threadIdx.x; i < rcount[j]; i += 256) dst[i] = __builtin_nontemporal_load(src + i);
}

// CG scalar step as its own launch: the reference-shaped (unfused) path, and after the
// all-reduce on several ranks (REDUCE = false: the sum is already in S->local).
template <int MODE, bool REDUCE>
__global__ __launch_bounds__(1024) void cg_scalar_k(uint32_t m, const double* __restrict__ q,
    CgScalars* S, double* __restrict__ rr_hist, double* __restrict__ pAp_hist, int to_local, int defer_x, int l1)
{
  __shared__ double lds16[16];
  // This launch sits on the critical path of every iteration: do not serialise the control block's
  // round trip in front of the partial loads -- fetch it, reduce (harmless if the loop has
  // already exited), and only then branch on it.
  const CgScalars in = cg_fetch(S);
  const int stopped  = in.stop;
  double total;
  if (REDUCE) {
    total = reduce_final_1024(m, q, lds16, l1);
    if (stopped) return;
    if (to_local) {
      if (threadIdx.x == 0) S->local = total;
      return;
    }
  } else {
    total = in.local;
    if (stopped) return;
  }
  if (threadIdx.x == 0) cg_apply<MODE>(S, in, total, rr_hist, pAp_hist, defer_x);
}

// The same step on several ranks with the all-reduce inside (p2p_allreduce_sum): local levels
// 1-2, exchange, step -- one launch.  (Its own kernel: the single-rank step above stays as it is.)
template <int MODE>
__global__ __launch_bounds__(1024) void cg_scalar_p2p_k(uint32_t m, const double* __restrict__ q,
    CgScalars* S, double* __restrict__ rr_hist, double* __restrict__ pAp_hist, int defer_x,
    const P2PView* __restrict__ pv, unsigned long long seq, int l1, const int* __restrict__ haloErr)
{
  __shared__ double lds16[16];
  const CgScalars in = cg_fetch(S);
  const int stopped  = in.stop; // identical on every rank: all of them skip the exchange, or none -- unless one has failed
  double total       = reduce_final_1024(m, q, lds16, l1);
  if (stopped) {
    if (in.p2p_error || (haloErr && *haloErr)) p2p_poison_allreduce(pv); // this rank has failed: nobody waits for its contribution
    return;
  }
  __syncthreads(); // lds16 is reused
  total = p2p_allreduce_sum(pv, total, seq, lds16, &S->p2p_error);
  // (atomic load: the line holding p2p_error was read with the control block above, a plain load could be served stale)
  if (__hip_atomic_load(&S->p2p_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // uniform: raised before the barriers inside the exchange
    if (threadIdx.x == 0) S->stop = 1;
    p2p_poison_allreduce(pv); // (and tell the others at once)
    return;
  }
  if (threadIdx.x == 0) cg_apply<MODE>(S, in, total, rr_hist, pAp_hist, defer_x);
}

// =============================================================================
// The vector phase of a CG body as ONE launch (cg_vector_phase_k): alpha | x, r update + r.r | beta, loop
// test | p update -- src/CGSolver.c:124-128 of body k and :107-116 of body k+1.  As separate launches these
// are four dependent kernels (two of them single-workgroup scalar steps); a kernel boundary on this part
// costs ~4 us (eight L2s to write back and invalidate), the four kernels move 134 MB, and r travels to memory
// and back between them.  Here every thread keeps its elements of r, p, x, Ap in registers across the
// two scalar steps; the steps themselves are taken by workgroup 0 while the others wait on a flag:
//   A  all: load r, Ap, p, x.   WG 0: levels 1-2 of p.Ap (partials written by the SpMV), [all-reduce],
//      alpha; publishes alpha.
//   B  all: r -= alpha Ap, x += alpha p (stored), level-0 partials of r.r (stored), then count in.
//   C  WG 0: once every workgroup has counted in: levels 1-2 of r.r, [all-reduce], beta and the loop test;
//      publishes beta and the stop flag.
//   D  all: p = r + beta p (unless the loop has ended).
// The arithmetic per element and the dot order are those of the separate kernels: same bits.
// What crosses workgroups inside the launch (partials, alpha, beta, the flags, the counter) travels by
// agent-scope relaxed atomics -- on this part they bypass the XCD-private L2s -- ordered by "all my stores are
// acknowledged" (s_waitcnt vmcnt(0)) in front of the flag / the count; no L2 write-back fence anywhere.
// The grid is sized so that every workgroup is resident (the waits would otherwise never end); all waits
// are bounded by a wall-clock timeout that raises VPhase::error (the host reports it at the end of the solve).
// =============================================================================
struct VPhase { // device control block, zeroed once
  unsigned long long arrived;  // workgroups that have finished phase B, over all launches
  unsigned long long launches; // finished launches (the next launch's sequence number - 1)
  unsigned long long flagA, alphaBits;
  unsigned long long flagB, betaBits, stopB;
  int error;
};

#ifdef SB_LAB // the one-launch vector phase and the lead kernels: measured slower than the five launches; lab builds only

__device__ __forceinline__ unsigned long long vp_load(const unsigned long long* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vp_store(unsigned long long* p, unsigned long long v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vp_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// thread 0 of the workgroup: wait until *flag >= want (bounded); false on timeout
__device__ __forceinline__ bool vp_wait(const unsigned long long* flag, unsigned long long want, long long timeoutTicks)
{
  const long long t0 = wall_clock64();
  while (vp_load(flag) < want) {
    if (wall_clock64() - t0 > timeoutTicks) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

// reduce_final_1024 on partials written during THIS launch by other workgroups (agent-scope loads)
__device__ __forceinline__ double reduce_final_1024_coherent(uint32_t m, const double* q, double* lds16)
{
  auto l1 = [&](uint32_t i) {
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(q) + 4u * (size_t)i;
    const double a = __longlong_as_double((long long)vp_load(w)), b = __longlong_as_double((long long)vp_load(w + 1));
    const double c = __longlong_as_double((long long)vp_load(w + 2)), d = __longlong_as_double((long long)vp_load(w + 3));
    return ((a + b) + c) + d;
  };
  double s   = 0.0;
  uint32_t i = threadIdx.x;
  for (; i + 3u * 1024u < m; i += 4u * 1024u) { // 16 loads in flight (the same sequence of additions as reduce_final_1024)
    double a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = l1(i + (uint32_t)u * 1024u);
#pragma unroll
    for (int u = 0; u < 4; u++) s = s + a[u];
  }
  for (; i < m; i += 1024u) s = s + l1(i);
  s = butterfly64(s);
  if ((threadIdx.x & 63u) == 0) lds16[threadIdx.x >> 6] = s;
  __syncthreads();
  double total = lds16[0];
#pragma unroll
  for (int w = 1; w < 16; w++) total = total + lds16[w];
  return total;
}

// SP: spans (128 consecutive elements, two per lane) a wave keeps in registers; the host picks the
// instantiation and a grid of resident workgroups with nSpans <= 16 * gridDim.x * SP.
template <int SP, bool P2P>
__global__ __launch_bounds__(1024) void cg_vector_phase_k(uint32_t n, double* r, double* p, const double* __restrict__ Ap,
    double* x, CgScalars* S, const double* __restrict__ pApPartials, double* rrPartials, uint32_t m,
    double* __restrict__ rr_hist, double* __restrict__ pAp_hist, VPhase* V, long long timeoutTicks, const P2PView* pv,
    unsigned long long p2pSeq, int pApL1)
{
  __shared__ double lds16[16];
  __shared__ double shVal;
  __shared__ int shFlag; // 0 go on, 1 loop ended, 2 error
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 16u + (threadIdx.x >> 6), nWaves = gridDim.x * 16u;
  const uint32_t nSpans = ((n + 255u) >> 8) * 2u;
  double2 rv[SP], av[SP], pw[SP], xv[SP];
#pragma unroll
  for (int k = 0; k < SP; k++) {
    const uint32_t e = (wave + (uint32_t)k * nWaves) * 128u + lane * 2u;
    rv[k] = av[k] = pw[k] = xv[k] = double2{ 0.0, 0.0 };
    if (e + 1 < n) {
      rv[k] = *reinterpret_cast<const double2*>(r + e), av[k] = *reinterpret_cast<const double2*>(Ap + e);
      pw[k] = *reinterpret_cast<const double2*>(p + e), xv[k] = *reinterpret_cast<const double2*>(x + e);
    } else if (e < n) {
      rv[k].x = r[e], av[k].x = Ap[e], pw[k].x = p[e], xv[k].x = x[e];
    }
  }
  const int stopped            = S->stop;
  const unsigned long long seq = V->launches + 1ull;
  if (stopped) return; // every workgroup sees the same flag: it only changes in phase C, after all have read it
  // ---- A: alpha -------------------------------------------------------------------------------------
  if (blockIdx.x == 0) {
    double total = reduce_final_1024(m, pApPartials, lds16, pApL1);
    __syncthreads();
    if (P2P) total = p2p_allreduce_sum(pv, total, p2pSeq, lds16, &S->p2p_error);
    if (threadIdx.x == 0) {
      cg_apply<2>(S, total, rr_hist, pAp_hist, 0);
      vp_store(&V->alphaBits, (unsigned long long)__double_as_longlong(S->alpha));
      vp_stores_done();
      vp_store(&V->flagA, seq);
    }
  }
  if (threadIdx.x == 0) {
    const bool ok = vp_wait(&V->flagA, seq, timeoutTicks);
    shVal  = __longlong_as_double((long long)vp_load(&V->alphaBits));
    shFlag = ok ? 0 : 2;
  }
  __syncthreads();
  if (shFlag == 2) { // (uniform per workgroup) a wait timed out: not every workgroup is running
    if (threadIdx.x == 0) atomicExch(&V->error, 1), S->stop = 1; // (stop: the bodies already enqueued return at once)
    return;
  }
  const double alpha = shVal, nalpha = -alpha;
  // ---- B: x, r, level 0 of r.r ------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < SP; k++) {
    const uint32_t sp = wave + (uint32_t)k * nWaves, e = sp * 128u + lane * 2u;
    double t = 0.0;
    if (e + 1 < n) {
      xv[k].x = xv[k].x + alpha * pw[k].x, xv[k].y = xv[k].y + alpha * pw[k].y;
      rv[k].x = rv[k].x + nalpha * av[k].x, rv[k].y = rv[k].y + nalpha * av[k].y;
      *reinterpret_cast<double2*>(x + e) = xv[k];
      *reinterpret_cast<double2*>(r + e) = rv[k];
      t = rv[k].x * rv[k].x + rv[k].y * rv[k].y;
    } else if (e < n) {
      xv[k].x = xv[k].x + alpha * pw[k].x;
      rv[k].x = rv[k].x + nalpha * av[k].x;
      x[e] = xv[k].x, r[e] = rv[k].x;
      t    = rv[k].x * rv[k].x + 0.0;
    }
    t = butterfly32(t);
    if (sp < nSpans && (lane & 31u) == 0)
      vp_store(reinterpret_cast<unsigned long long*>(rrPartials) + sp * 2u + (lane >> 5), (unsigned long long)__double_as_longlong(t));
  }
  vp_stores_done();
  __syncthreads(); // (also: nobody overwrites shVal / shFlag before everybody has read them)
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&V->arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- C: beta, loop test ------------------------------------------------------------------------------
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) shFlag = vp_wait(&V->arrived, (unsigned long long)gridDim.x * seq, timeoutTicks) ? 0 : 2;
    __syncthreads();
    if (shFlag != 2) {
      double total = reduce_final_1024_coherent(m, rrPartials, lds16);
      __syncthreads();
      if (P2P) total = p2p_allreduce_sum(pv, total, p2pSeq + 1ull, lds16, &S->p2p_error);
      if (threadIdx.x == 0) {
        cg_apply<1>(S, total, rr_hist, pAp_hist, 0);
        if (P2P && __hip_atomic_load(&S->p2p_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) S->stop = 1;
        vp_store(&V->betaBits, (unsigned long long)__double_as_longlong(S->beta));
        vp_store(&V->stopB, (unsigned long long)S->stop);
        vp_stores_done();
        vp_store(&V->flagB, seq);
        V->launches = seq;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool ok = vp_wait(&V->flagB, seq, timeoutTicks);
    shVal  = __longlong_as_double((long long)vp_load(&V->betaBits));
    shFlag = !ok ? 2 : vp_load(&V->stopB) ? 1 : 0;
  }
  __syncthreads();
  if (shFlag == 2) {
    if (threadIdx.x == 0) atomicExch(&V->error, 1), S->stop = 1; // (stop: the bodies already enqueued return at once)
    return;
  }
  if (shFlag == 1) return; // the loop has ended: p stays (the separate p update returns on the stop flag too)
  // ---- D: p = r + beta p ---------------------------------------------------------------------------------
  const double beta = shVal;
#pragma unroll
  for (int k = 0; k < SP; k++) {
    const uint32_t e = (wave + (uint32_t)k * nWaves) * 128u + lane * 2u;
    if (e + 1 < n) {
      double2 o;
      o.x = rv[k].x + beta * pw[k].x, o.y = rv[k].y + beta * pw[k].y;
      *reinterpret_cast<double2*>(p + e) = o;
    } else if (e < n) {
      p[e] = rv[k].x + beta * pw[k].x;
    }
  }
}

#endif // SB_LAB (vector phase)

// =============================================================================
// Scalar steps inside their consumers ("lead" kernels): 5 -> 3 launches per CG body.
// (One rank only: with several ranks the combination with the in-kernel all-reduce and the in-SpMV halo wait timed out
// in the two-ranks-on-one-GPU test and, being slower anyway, was not pursued.)
// The alpha step (levels 1-2 of p.Ap, alpha = rr / pAp) needs only the partials the SpMV has
// written, and its only consumers are the kernel that updates r -- so workgroup 0 of THAT kernel takes the step
// while the other workgroups already have their first loads in flight, then publishes alpha through a flag
// (agent-scope relaxed atomics, vp_* above) on which the others wait.  The same for the beta step / loop test in
// front of the p update.  Unlike cg_vector_phase_k nobody waits for ALL workgroups, only for workgroup 0, which is
// dispatched first: no residency requirement (the launch counter moves when the LAST workgroup leaves, lead_leave), the
// reads and writes of the kernel still overlap freely, and the
// ~4 us of a dependent single-workgroup launch become the ~2 us the reduction itself takes.
// 1024 threads per workgroup, so that workgroup 0 IS the reduction workgroup of the canonical dot.
// =============================================================================
struct Lead { // device control of one lead kernel, zeroed once
  unsigned long long flag, valueBits, stop;
  unsigned long long launches; // finished launches: the next one's sequence number - 1
  unsigned long long done;     // workgroups of the running launch that have read `launches` and left (lead_leave)
  int error;
};

#ifdef SB_LAB

// Every workgroup reads Ld->launches when it starts, and a grid of more workgroups than the device holds starts in
// rounds: the counter may therefore only move once EVERY workgroup of the launch has read it.  The last workgroup to
// leave advances it (round 2 had workgroup 0 do so in the middle of the launch: a workgroup dispatched after that
// read the new value, waited for a flag nobody would publish and ran into the time-out -- ADVICE r2).
__device__ __forceinline__ void lead_leave(Lead* Ld, unsigned long long seq)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long prev = __hip_atomic_fetch_add(&Ld->done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned long long)gridDim.x - 1ull) {
      vp_store(&Ld->done, 0ull);
      vp_store(&Ld->launches, seq);
    }
  }
}

// workgroup 0: total = the finished dot product.  Everybody returns the published (value, stop); false: timeout
template <int MODE>
__device__ __forceinline__ bool lead_step(CgScalars* S, const double* __restrict__ partials, uint32_t m,
    double* __restrict__ rr_hist, double* __restrict__ pAp_hist, Lead* Ld, unsigned long long seq, long long timeoutTicks,
    double* lds16, double* shVal, int* shFlag, double& value, int& stop, int l1 = 0)
{
  if (blockIdx.x == 0) {
    const double total = reduce_final_1024(m, partials, lds16, l1);
    if (threadIdx.x == 0) {
      cg_apply<MODE>(S, total, rr_hist, pAp_hist, 1);
      vp_store(&Ld->valueBits, (unsigned long long)__double_as_longlong(MODE == 2 ? S->alpha : S->beta));
      vp_store(&Ld->stop, (unsigned long long)S->stop);
      vp_stores_done();
      vp_store(&Ld->flag, seq);
    }
  }
  if (threadIdx.x == 0) {
    const bool ok = vp_wait(&Ld->flag, seq, timeoutTicks);
    *shVal  = __longlong_as_double((long long)vp_load(&Ld->valueBits));
    *shFlag = !ok ? 2 : vp_load(&Ld->stop) ? 1 : 0;
    if (!ok) atomicExch(&Ld->error, 1), S->stop = 1;
  }
  __syncthreads();
  value = *shVal, stop = *shFlag;
  return *shFlag != 2;
}

// alpha step + r -= alpha Ap + level-0 partials of r.r   (src/CGSolver.c:124-126, :128, :112)
// = cg_scalar_k<2> followed by the r update (cg_update_r_k's arithmetic, level-0 partials), element for element
__global__ __launch_bounds__(1024) void cg_lead_r_k(uint32_t n, const double* __restrict__ Ap, double* r, CgScalars* S,
    const double* __restrict__ pApPartials, double* __restrict__ rrPartials, uint32_t m, double* __restrict__ rr_hist,
    double* __restrict__ pAp_hist, Lead* Ld, long long timeoutTicks, int pApL1)
{
  __shared__ double lds16[16];
  __shared__ double shVal;
  __shared__ int shFlag;
  const uint32_t lane   = threadIdx.x & 63u;
  const uint32_t nSpans = ((n + 255u) >> 8) * 2u;
  const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
  uint32_t s            = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // the first two spans' loads go in flight in front of the step
  const bool pair0 = s + nWaves < nSpans && (s + nWaves) * 128u + 128u <= n; // wave-uniform
  double2 r0 = { 0.0, 0.0 }, a0 = r0, r1 = r0, a1 = r0;
  if (pair0) {
    const uint32_t e0 = s * 128u + lane * 2u, e1 = (s + nWaves) * 128u + lane * 2u;
    r0 = *reinterpret_cast<const double2*>(r + e0), a0 = *reinterpret_cast<const double2*>(Ap + e0);
    r1 = *reinterpret_cast<const double2*>(r + e1), a1 = *reinterpret_cast<const double2*>(Ap + e1);
  }
  const int stopped            = S->stop;
  const unsigned long long seq = vp_load(&Ld->launches) + 1ull;
  if (stopped) return; // (the same decision in every workgroup: nobody counts, the counter stays)
  double alpha;
  int st;
  if (!lead_step<2>(S, pApPartials, m, rr_hist, pAp_hist, Ld, seq, timeoutTicks, lds16, &shVal, &shFlag, alpha, st, pApL1)) return; // (fatal at the host)
  const double nalpha = -alpha;
  bool have = pair0;
  while (have) {
    const uint32_t e0 = s * 128u + lane * 2u, e1 = (s + nWaves) * 128u + lane * 2u;
    r0.x = r0.x + nalpha * a0.x, r0.y = r0.y + nalpha * a0.y;
    r1.x = r1.x + nalpha * a1.x, r1.y = r1.y + nalpha * a1.y;
    *reinterpret_cast<double2*>(r + e0) = r0;
    *reinterpret_cast<double2*>(r + e1) = r1;
    const double t0 = butterfly32(r0.x * r0.x + r0.y * r0.y), t1 = butterfly32(r1.x * r1.x + r1.y * r1.y);
    if ((lane & 31u) == 0) rrPartials[s * 2u + (lane >> 5)] = t0, rrPartials[(s + nWaves) * 2u + (lane >> 5)] = t1;
    s += 2u * nWaves;
    have = s + nWaves < nSpans && (s + nWaves) * 128u + 128u <= n;
    if (have) {
      const uint32_t f0 = s * 128u + lane * 2u, f1 = (s + nWaves) * 128u + lane * 2u;
      r0 = *reinterpret_cast<const double2*>(r + f0), a0 = *reinterpret_cast<const double2*>(Ap + f0);
      r1 = *reinterpret_cast<const double2*>(r + f1), a1 = *reinterpret_cast<const double2*>(Ap + f1);
    }
  }
  for (; s < nSpans; s += nWaves) { // what the paired loop left over
    const uint32_t e = s * 128u + lane * 2u;
    double t         = 0.0;
    if (e + 1 < n) {
      double2 rv       = *reinterpret_cast<double2*>(r + e);
      const double2 av = *reinterpret_cast<const double2*>(Ap + e);
      rv.x = rv.x + nalpha * av.x;
      rv.y = rv.y + nalpha * av.y;
      *reinterpret_cast<double2*>(r + e) = rv;
      t = rv.x * rv.x + rv.y * rv.y;
    } else if (e < n) {
      const double rn = r[e] + nalpha * Ap[e];
      r[e]            = rn;
      t               = rn * rn + 0.0;
    }
    t = butterfly32(t);
    if ((lane & 31u) == 0) rrPartials[s * 2u + (lane >> 5)] = t;
  }
  lead_leave(Ld, seq);
}

// beta step / loop test + p = r + beta p + the x update the previous body owes   (:107-116, :127)
// = cg_scalar_k<1> (defer_x) followed by cg_update_p(which = 0, x), element for element
__global__ __launch_bounds__(1024) void cg_lead_p_k(uint32_t n, const double* __restrict__ r, double* p, double* x,
    CgScalars* S, const double* __restrict__ rrPartials, uint32_t m, double* __restrict__ rr_hist,
    double* __restrict__ pAp_hist, Lead* Ld, long long timeoutTicks)
{
  __shared__ double lds16[16];
  __shared__ double shVal;
  __shared__ int shFlag;
  const uint32_t n2     = n >> 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const double2* r2     = reinterpret_cast<const double2*>(r);
  double2* p2           = reinterpret_cast<double2*>(p);
  double2* x2           = reinterpret_cast<double2*>(x);
  uint32_t i            = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t last   = n2 ? n2 - 1u : 0u;
  double2 a0 = { 0.0, 0.0 }, b0 = a0, x0 = a0, a1 = a0, b1 = a0, x1 = a0;
  auto load = [&](uint32_t j, double2& a, double2& b, double2& xv) { a = r2[j], b = p2[j], xv = x2[j]; };
  if (n2) load(min(i, last), a0, b0, x0), load(min(i + stride, last), a1, b1, x1);
  const int stopped            = S->stop;
  const double alpha           = S->alpha; // of the previous body: written by the kernel before this one
  const unsigned long long seq = vp_load(&Ld->launches) + 1ull;
  if (stopped) return;
  double beta;
  int st;
  if (!lead_step<1>(S, rrPartials, m, rr_hist, pAp_hist, Ld, seq, timeoutTicks, lds16, &shVal, &shFlag, beta, st)) return;
  if (st) { // the loop has ended: p stays, the x update stays owed (cg_x_finalize)
    lead_leave(Ld, seq);
    return;
  }
  auto finish = [&](uint32_t j, const double2& a, const double2& b, double2 xv) {
    xv.x = xv.x + alpha * b.x;
    xv.y = xv.y + alpha * b.y;
    x2[j] = xv;
    double2 o;
    o.x = a.x + beta * b.x;
    o.y = a.y + beta * b.y;
    p2[j] = o;
  };
  for (; i < n2; i += 2u * stride) {
    const bool second = i + stride < n2;
    finish(i, a0, b0, x0);
    if (second) finish(i + stride, a1, b1, x1);
    const uint32_t nx = i + 2u * stride;
    if (nx < n2) load(nx, a0, b0, x0), load(min(nx + stride, last), a1, b1, x1);
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double bb = p[n - 1];
    x[n - 1]        = x[n - 1] + alpha * bb;
    p[n - 1]        = r[n - 1] + beta * bb;
  }
  lead_leave(Ld, seq);
}

#endif // SB_LAB (lead kernels)

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
