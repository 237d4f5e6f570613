// sbhip_launch.inc.h -- part of the single translation unit sbhip.hip (textual include, shares its
// static context): kernel launches: SpMV (all formats / modes), waxpby, dots, permutation helpers.
// ===========================================================================
// kernels
// ===========================================================================
// SpMV launch timing (sb_cg_spmv_timing; bench.py's roofline leg): while the CG loop has set a pair of events, the SpMV
// launch below goes through hipExtLaunchKernelGGL, which stamps them with the BEGIN and END of that kernel's own
// execution -- the launch duration rocprofv3 reports -- instead of bracketing the launch with hipEventRecord, which also
// counts the gap to the previous kernel's end and the dispatch (~5-14 us on a kernel that follows a vector kernel).
static hipEvent_t g_spmvEvA = nullptr, g_spmvEvB = nullptr;
#define SB_SPMV_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                  \
  do {                                                                                                           \
    if (g_spmvEvA) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, g_spmvEvA, g_spmvEvB, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                    \
  } while (0)

static int g_scs_unroll = -1;
static int g_scs_nt     = -1;
static int g_scs_xcd    = 1;

// dotPartials != NULL: fuse the level-0 partials of p.Ap into the SpMV (SCS C=64 only)
// part: 0 the whole product; 1 / 2 its interior / halo-touching tiles (spmv_can_split only)
static bool spmv_uses_patterns(const sb_matrix* m)
{
  return (m->usePacked == 3 || m->usePacked == 5) && (m->fmt == 0 ? m->mirror != nullptr : m->C == 64);
}
static bool spmv_can_split(const sb_matrix* m)
{
  const sb_matrix* pm = pat_of(m);
  if (!spmv_uses_patterns(m)) return false;
  if (m->usePacked == 5) return pm->mInterior > 0 && pm->mInterior < pm->mNTiles;
  return pm->patInterior > 0 && pm->patInterior < pm->patNTiles;
}
static void launch_pat(const sb_matrix* pm, bool skipPad, bool masked, const double* x, double* y, double* dotPartials,
    const int* stop, int part, hipStream_t stream, const HaloWait* halo);

// halo != NULL (pattern kernel only): the halo-touching tiles wait for the neighbours' pushes themselves
static void launch_spmv(const sb_matrix* m, const double* x, double* y, double* dotPartials,
    const int* stop, int part = 0, hipStream_t stream = nullptr, const HaloWait* halo = nullptr)
{
  const bool dot = dotPartials != nullptr;
  if (m->nr == 0) return;
  if (part != 0 && !spmv_can_split(m)) SB_FATAL("this SpMV kernel cannot be launched in parts");
  if (m->fmt == 0 && spmv_uses_patterns(m)) {
    launch_pat(m->mirror, true, m->usePacked == 5, x, y, dotPartials, stop, part, stream ? stream : g.stream, halo);
  } else if (m->fmt == 0) {
    if (dot) SB_FATAL("the native CRS kernel has no fused dot (its row blocks are not aligned to the 64-row groups of "
                      "the canonical dot; a kernel that is was measured slower, kernels.hip.h): sbhip_cg adds a dot pass");
    if (m->tileRow) {
      const uint32_t per = (m->nCrsTiles + 7) / 8;
      SB_SPMV_LAUNCH(spmv_crs_split, dim3(per * 8), dim3(CRS_THREADS), 0, g.stream, m->tileRow, m->rowPtr, m->colInd, m->val, x, y,
          m->nCrsTiles, m->crsT, m->nnz, per, stop);
      HIP_CHECK(hipGetLastError());
      return;
    }
    const uint32_t per = (m->nRowBlocks + 7) / 8;
    SB_SPMV_LAUNCH(spmv_crs_stream, dim3(per * 8), dim3(CRS_THREADS), 0, g.stream, m->rowBlocks, m->rowPtr, m->colInd,
        m->val, x, y, m->nRowBlocks, per, stop);
  } else if (m->C == 64) {
    if (g_scs_unroll < 0) {
      const char* u = getenv("SB_SCS_UNROLL");
      g_scs_unroll  = u ? atoi(u) : 4;
      const char* n = getenv("SB_SCS_NT");
      g_scs_nt      = n ? atoi(n) : 1;
      const char* xc = getenv("SB_SCS_XCD");
      g_scs_xcd     = xc ? atoi(xc) : 1;
    }
    const uint32_t nBlocks = (m->nChunks + 3) / 4;
    const uint32_t per     = g_scs_xcd ? (nBlocks + 7) / 8 : 0;
    dim3 grid(g_scs_xcd ? per * 8 : nBlocks), block(256);
    if (m->usePacked == 3 || m->usePacked == 5) {
      launch_pat(m, false, m->usePacked == 5, x, y, dotPartials, stop, part, stream ? stream : g.stream, halo);
#ifdef SB_LAB // levels 1-3 of the compressed mirror: measured slower than level 6 at every size; lab builds only
    } else if (m->usePacked == 2) {
      const size_t shmem = (256 + (size_t)m->ldsWindow) * sizeof(double);
#define LDS_LAUNCH(DI, DO)                                                                                   \
  SB_SPMV_LAUNCH((spmv_scs64_lds<DI, DO>), grid, block, shmem, g.stream, m->pmeta, m->pslots, m->pcodes, \
      m->pdict, m->chunkPtr, m->val, m->tileSegPtr, m->tileSegs, x, y, m->nr, m->nChunks, per, m->padCol,    \
      dotPartials, stop)
      if (m->nDict > 0) {
        if (dot) LDS_LAUNCH(true, true);
        else LDS_LAUNCH(true, false);
      } else {
        if (dot) LDS_LAUNCH(false, true);
        else LDS_LAUNCH(false, false);
      }
#undef LDS_LAUNCH
    } else if (m->usePacked == 1) {
#define PK_LAUNCH(DI, DO)                                                                                 \
  SB_SPMV_LAUNCH((spmv_scs64_packed<DI, DO>), grid, block, 0, g.stream, m->pmeta, m->pidx, m->pcodes, \
      m->pdict, m->chunkPtr, m->val, x, y, m->nr, m->nChunks, per, m->padCol, dotPartials, stop)
      if (m->nDict > 0) {
        if (dot) PK_LAUNCH(true, true);
        else PK_LAUNCH(true, false);
      } else {
        if (dot) PK_LAUNCH(false, true);
        else PK_LAUNCH(false, false);
      }
#undef PK_LAUNCH
#endif // SB_LAB
    } else {
#define SCS_LAUNCH(U, D, N)                                                                      \
  SB_SPMV_LAUNCH((spmv_scs64<U, D, N>), grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, \
      m->colInd, m->val, x, y, m->nr, m->nChunks, per, dotPartials, stop)
#define SCS_PICK(U)                                                                           \
  do {                                                                                        \
    if (dot) { if (g_scs_nt) SCS_LAUNCH(U, true, true); else SCS_LAUNCH(U, true, false); }    \
    else { if (g_scs_nt) SCS_LAUNCH(U, false, true); else SCS_LAUNCH(U, false, false); }      \
  } while (0)
#ifdef SB_LAB // unroll depths other than 4: measured equal or slower
      switch (g_scs_unroll) {
      case 1: SCS_PICK(1); break;
      case 2: SCS_PICK(2); break;
      case 8: SCS_PICK(8); break;
      case 9: SCS_PICK(9); break;
      default: SCS_PICK(4); break;
      }
#else
      SCS_PICK(4);
#endif
#undef SCS_PICK
#undef SCS_LAUNCH
    }
  } else {
    if (dot) SB_FATAL("fused dot is an SCS C=64 feature");
    SB_SPMV_LAUNCH(spmv_scs_generic, dim3((m->nrPadded + 255) / 256), dim3(256), 0, g.stream,
        m->chunkPtr, m->chunkLens, m->colInd, m->val, x, y, m->nr, m->nrPadded, m->C, stop);
  }
  HIP_CHECK(hipGetLastError());
}

// masked: the level-6 form of the same tiles (row programs, pack.hip.h)
static void launch_pat(const sb_matrix* pm, bool skipPad, bool masked, const double* x, double* y, double* dotPartials,
    const int* stop, int part, hipStream_t stream, const HaloWait* halo)
{
  HaloWait hw;
  memset(&hw, 0, sizeof hw);
  if (halo) hw = *halo;
  if (masked && !pm->mHdrs) SB_FATAL("the matrix has no masked row programs");
  const bool dot         = dotPartials != nullptr;
  const uint32_t nBlocks = masked ? pm->mNTiles : pm->patNTiles, interior = masked ? pm->mInterior : pm->patInterior;
  const uint32_t dictE = masked ? pm->mDict : pm->patDict, excE = masked ? 0u : pm->patExcLds;
  const size_t shmem = ((size_t)dictE + excE + 8) * sizeof(PatEntry) + (size_t)(masked ? pm->mWindow : pm->patWindow) * sizeof(double);
  if (!stop) stop = zero_flag();
  const uint32_t first = part == 2 ? interior : 0u;
  const uint32_t count = part == 1 ? interior : part == 2 ? nBlocks - interior : nBlocks;
  const uint32_t pper  = g_scs_xcd ? (count + 7) / 8 : 0;
  const dim3 pgrid((g_scs_xcd ? pper * 8 : count) + (halo ? hw.nPush : 0u)), block(256);
  const uint32_t* hdrs   = masked ? pm->mHdrs : pm->tileHdrs;
  const uint32_t* codes  = masked ? pm->mStream : pm->jcodes;
  const uint16_t* rbase  = masked ? reinterpret_cast<const uint16_t*>(pm->mRowBase) : pm->rowBase;
#define PAT_LAUNCH(CP, DO, SK, HA, MA)                                                                                     \
  SB_SPMV_LAUNCH((spmv_scs64_pat<CP, DO, SK, HA, MA>), pgrid, block, shmem, stream, hdrs, codes, rbase,                \
      masked ? pm->mClassDict : pm->classDict, pm->rowPats, pm->excRows, pm->mProgs, pm->mSlotMap, pm->mMapStride,           \
      masked ? pm->mSegs : pm->patSegs, x, y, pm->nr, pm->nChunks, first, count, pper, pm->padCol, dictE, excE, dotPartials, \
      stop, hw)
#define PAT_PICK(CP, SK, HA, MA)                \
  do {                                          \
    if (dot) PAT_LAUNCH(CP, true, SK, HA, MA);  \
    else PAT_LAUNCH(CP, false, SK, HA, MA);     \
  } while (0)
#define PAT_PICK2(CP, MA)                       \
  do {                                          \
    if (skipPad) {                              \
      if (halo) PAT_PICK(CP, true, true, MA);   \
      else PAT_PICK(CP, true, false, MA);       \
    } else {                                    \
      if (halo) PAT_PICK(CP, false, true, MA);  \
      else PAT_PICK(CP, false, false, MA);      \
    }                                           \
  } while (0)
#ifndef SB_LAB // the product ships the masked row programs (level 6) only; levels 4-5 are lab builds
  if (!masked) SB_FATAL("levels 4-5 of the compressed mirror are compiled into lab builds only (-DSB_LAB)");
#endif
  if ((masked ? pm->mCPT : pm->patCPT) == 8) {
    if (masked) PAT_PICK2(8, true);
#ifdef SB_LAB
    else PAT_PICK2(8, false);
#endif
  } else {
    if (masked) PAT_PICK2(4, true);
#ifdef SB_LAB
    else PAT_PICK2(4, false);
#endif
  }
#undef PAT_PICK2
#undef PAT_PICK
#undef PAT_LAUNCH
  HIP_CHECK(hipGetLastError());
}

// The SpMV that takes the p update (pack.hip.h: spmv_prog_fusep): the masked row programs are the selected kernel, EVERY
// chunk is a row program (no per-lane code words) and every window is of the mapped or of the simple kind.
static bool spmv_fusep_possible(const sb_matrix* m)
{
  if (m->usePacked != 5 || !spmv_uses_patterns(m)) return false;
  const sb_matrix* pm = pat_of(m);
  return pm->mHdrs && pm->mDict == 0 && pm->nMaskedChunks == pm->nChunks && (pm->mSlotMap != nullptr || pm->mAllSimple);
}
// Ap = A p_new with p_new = r + beta p_old formed on the way (which != 0: the first body, p_new = r + 0.0 * r), x += alpha p_old
// where the previous body owes it, level-1 values of p_new . Ap into dotL1
static void launch_spmv_fusep(const sb_matrix* m, const double* pold, const double* r, double* pnew, double* xsol, double* y,
    const CgScalars* S, int which, double* dotL1, const HaloWait* halo)
{
  const sb_matrix* pm = pat_of(m);
  const bool skipPad = m->fmt == 0, mapped = pm->mSlotMap != nullptr;
  HaloWait hw;
  memset(&hw, 0, sizeof hw);
  if (halo) hw = *halo;
  const uint32_t count = pm->mNTiles;
  const uint32_t pper  = g_scs_xcd ? (count + 7) / 8 : 0;
  const dim3 pgrid((g_scs_xcd ? pper * 8 : count) + (halo ? hw.nPush : 0u)), block(256);
  const size_t shmem = (16 + (size_t)pm->mWindow) * sizeof(double);
#define FP_LAUNCH(CP, SK, HA, MP)                                                                                          \
  SB_SPMV_LAUNCH((spmv_prog_fusep<CP, SK, HA, MP>), pgrid, block, shmem, g.stream, pm->mHdrs, pm->mRowBase, pm->mProgs,  \
      pm->mSlotMap, pm->mMapStride, pold, r, pnew, xsol, y, S, which, pm->nr, pm->nChunks, 0u, count, pper, pm->padCol, dotL1, hw)
#define FP_PICK(CP, SK, HA)               \
  do {                                    \
    if (mapped) FP_LAUNCH(CP, SK, HA, true); \
    else FP_LAUNCH(CP, SK, HA, false);    \
  } while (0)
#define FP_PICK2(CP)                      \
  do {                                    \
    if (skipPad) {                        \
      if (halo) FP_PICK(CP, true, true);  \
      else FP_PICK(CP, true, false);      \
    } else {                              \
      if (halo) FP_PICK(CP, false, true); \
      else FP_PICK(CP, false, false);     \
    }                                     \
  } while (0)
  if (pm->mCPT == 8) FP_PICK2(8);
  else FP_PICK2(4);
#undef FP_PICK2
#undef FP_PICK
#undef FP_LAUNCH
  HIP_CHECK(hipGetLastError());
}

void sb_spmv_native(const sb_matrix* m, const double* x, double* y)
{
  need_init();
  launch_spmv(m, x, y, nullptr, nullptr);
}

// which values the fused dot of the selected SpMV kernel writes: 0 none (no fused dot), 1 level-0 partials (one per 64
// rows), 2 LEVEL-1 values (one per 256 rows: the product's two wave-per-chunk kernels combine a block's / tile's four
// chunks themselves; the lab-only kernels keep level 0)
static int spmv_dot_kind(const sb_matrix* m)
{
  if (!(m->fmt == 1 ? m->C == 64 : spmv_uses_patterns(m))) return 0;
  if (m->usePacked == 5 || (m->fmt == 1 && m->usePacked == 0)) return 2;
  return 1;
}

int sb_spmv_native_dot(const sb_matrix* m, const double* x, double* y, double* partials_dev)
{ // the product with the fused partials of x . y, as the CG loop launches it for p . Ap
  need_init();
  const int kind = spmv_dot_kind(m);
  if (!kind) return 0;
  launch_spmv(m, x, y, partials_dev, nullptr);
  return kind;
}

void sb_permute(const sb_matrix* m, const double* in_orig, double* out_perm)
{
  need_init();
  if (!m->permuted) {
    if (in_orig != out_perm) sb_d2d(out_perm, in_orig, (size_t)m->nr * sizeof(double));
    return;
  }
  hipLaunchKernelGGL(gather_k, dim3(stream_grid(m->nr, 256)), dim3(256), 0, g.stream, m->nr,
      m->newToOld, in_orig, out_perm, (const int*)nullptr);
  HIP_CHECK(hipGetLastError());
}

void sb_unpermute(const sb_matrix* m, const double* in_perm, double* out_orig)
{
  need_init();
  if (!m->permuted) {
    if (in_perm != out_orig) sb_d2d(out_orig, in_perm, (size_t)m->nr * sizeof(double));
    return;
  }
  hipLaunchKernelGGL(gather_k, dim3(stream_grid(m->nr, 256)), dim3(256), 0, g.stream, m->nr,
      m->oldToNew, in_perm, out_orig, (const int*)nullptr);
  HIP_CHECK(hipGetLastError());
}

void sb_spmv(const sb_matrix* m, const double* x, double* y)
{
  need_init();
  if (!m->permuted) {
    launch_spmv(m, x, y, nullptr, nullptr);
    return;
  }
  double* xp = scratch_ws(0, m->nc);
  double* yp = scratch_ws(1, m->nr);
  sb_permute(m, x, xp);
  if (m->nc > m->nr)
    sb_d2d(xp + m->nr, x + m->nr, (size_t)(m->nc - m->nr) * sizeof(double));
  launch_spmv(m, xp, yp, nullptr, nullptr);
  sb_unpermute(m, yp, y);
}

static void launch_waxpby(uint32_t n, double alpha, const double* x, double beta, const double* y,
    double* w, const int* stop)
{
  if (n == 0) return;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15u) SB_FATAL("waxpby: vectors must be 16-byte aligned");
  hipLaunchKernelGGL(waxpby_k, dim3(stream_grid(n / 2 + 1, 256)), dim3(256), 0, g.stream, n, alpha, x,
      beta, y, w, stop);
  HIP_CHECK(hipGetLastError());
}

void sb_waxpby(uint32_t n, double alpha, const double* x, double beta, const double* y, double* w)
{
  need_init();
  launch_waxpby(n, alpha, x, beta, y, w, nullptr);
}

// OP 0 dot(a,b) / OP 1 x,r update + r.r / OP 2 r = b - Ap + r.r  (kernels.hip.h: dot_spans_k);
// partials receives 4*ceil(n/256) level-0 partials (tail zeroed)
static void launch_dot_spans(int op, uint32_t n, const double* a, const double* b, double* x, double* r,
    const CgScalars* S, double* partials, const int* stop)
{
  if (n == 0) return;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)x | (uintptr_t)r) & 15u) SB_FATAL("vectors must be 16-byte aligned");
  const uint32_t nSpans = ((n + 255u) / 256u) * 2u;
  const dim3 grid(stream_grid(nSpans, 4)), block(256);
  if (op == 0) hipLaunchKernelGGL((dot_spans_k<0>), grid, block, 0, g.stream, n, a, b, x, r, S, partials, stop);
  else if (op == 1) hipLaunchKernelGGL((dot_spans_k<1>), grid, block, 0, g.stream, n, a, b, x, r, S, partials, stop);
  else if (op == 2) hipLaunchKernelGGL((dot_spans_k<2>), grid, block, 0, g.stream, n, a, b, x, r, S, partials, stop);
  else SB_FATAL("launch_dot_spans: unknown op %d", op);
  HIP_CHECK(hipGetLastError());
}

void sb_ddot_partials(uint32_t n, const double* x, const double* y, double* partials_dev)
{
  need_init();
  launch_dot_spans(0, n, x, y, nullptr, nullptr, nullptr, partials_dev, nullptr);
}

void sb_reduce_final(uint32_t m, const double* partials_dev, double* result_dev)
{
  need_init();
  hipLaunchKernelGGL(reduce_final_k, dim3(1), dim3(1024), 0, g.stream, m, partials_dev, result_dev,
      (const int*)nullptr);
  HIP_CHECK(hipGetLastError());
}

void sb_ddot_async(uint32_t n, const double* x, const double* y, double* result_dev)
{
  need_init();
  const uint32_t m = (n + 255u) / 256u;
  double* q        = scratch_partials(4 * (size_t)m);
  sb_ddot_partials(n, x, y, q);
  sb_reduce_final(m, q, result_dev);
  if (multi_rank()) sb_comm_reduction(result_dev, 1);
}

double sb_ddot(uint32_t n, const double* x, const double* y)
{
  need_init();
  sb_ddot_async(n, x, y, g.scalar);
  double r = 0.0;
  sb_d2h(&r, g.scalar, sizeof r);
  return r;
}

// ===========================================================================
// placement tuner (sbhip_matrix.inc.h: "placement of the reference-layout stream")
// ===========================================================================
// WHICH device memory the stream and the loop's vectors sit in decides how fast the section-8d loop runs: the same arrays copied
// into a series of fresh allocations of one process ran at 128 / 128 / 114 / 115 ... us per SpMV launch at HPCG 128^3, the CG
// step moves between 142 and 157 us with the allocation the vectors got (identical virtual addresses included), and the two
// interact (profiles/r04_placement_lab*.txt) -- while the position INSIDE an allocation changes nothing (289 offsets:
// 126.7-128.7 us).  Nothing a process can see (address, size, alignment) tells the kinds apart, so the upload MEASURES, with a
// proxy of the loop body on the loop's own vector layout (p = r + 0.5 p | Ap = A p | r = r - 1e-3 Ap: no scalars, no
// communication), in up to SB_PLACE_ROUNDS (6) rounds:
//   - SB_PLACE_VEC_TRIES (10) fresh allocations of the vectors' arena, each followed by a 700 MB spacer (pairs whose two
//     allocations lie within ~2.5 GB of each other are of the slowest kind), timed with the stream in the home it has;
//   - the stream copied into SB_PLACE_TRIES (4) fresh slabs, each timed with up to four arenas (the best so far, the newest, two
//     older ones): the level is a table over where BOTH allocations lie (every pair of 8 stream x 16 arena allocations of one
//     process: profiles/r04_placement_lab7.txt), so a new home for the stream may want another arena.  The first slab of the
//     first round is MANAGED memory preferred on and prefetched to this device (SB_PLACE_MANAGED=0: not): wherever the home
//     stream was slow with every plain arena, that slab was fast with every one of them (profiles/r04_placement_lab12.txt).
// Everything tried stays allocated until the end (the next try lands further on).  Pairs come in levels -- 131 | 137-139 | 143 |
// 147 | 153-157 us per proxy step at HPCG 128^3 -- in long runs along the allocation order.  The search ends once the pair kept is
// >= 14.5 % faster than the slowest pair seen or moves the proxy step's algorithmic bytes at >= 6.0 TB/s (the fast level), or after
// three rounds in a row without anything better (one device in three of the pool has no fast pairs at all: 104 probes between
// 151 and 157 us).  Physically contiguous allocations (hipDeviceMallocContiguous) and
// power-of-two sizes behave the same, and stream + vectors at 256 position pairs inside ONE 48 GiB allocation are all of the
// slow kind: it is neither fragmentation nor page-table fragment size, and it needs two allocations
// (profiles/r04_placement_probe_sequences.txt, r04_placement_lab6.txt).  The mechanism is not understood.  The fastest pair is
// kept -- the arena stays with the matrix and the next sb_cg_create takes its vectors from it -- the rest is freed.  0.05-0.4 s per
// upload and, for a moment, up to a quarter of the device's memory (less where less is free).  SB_PLACE=0 switches it off;
// streams below 64 MB (cache resident) are left alone.  Same bytes, same kernels, same arithmetic: same bits.
struct VecLayout {
  size_t r, Ap, x, b, p, p2, xexact, total;
};
static size_t vec_up(size_t v, size_t pad) { return ((v + 4095) & ~(size_t)4095) + pad; }
static VecLayout vec_layout(uint32_t nr, uint32_t nc, bool hasExact, size_t pad)
{
  const size_t nb = (size_t)nr * sizeof(double), nbc = (size_t)nc * sizeof(double);
  VecLayout L;
  size_t off = 0;
  auto take = [&](size_t v) { const size_t o = off; off += vec_up(v, pad); return o; };
  L.r = take(nb), L.Ap = take(nb), L.x = take(nb), L.b = take(nb), L.p = take(nbc), L.p2 = take(nbc);
  L.xexact = hasExact ? take(nb) : 0;
  L.total  = off + 4096;
  return L;
}
// one proxy step on vectors at `arena` (laid out as the loop lays them out), best of two batches of four
static float placement_probe(sb_matrix* m, char* arena, const VecLayout& L, hipEvent_t ea, hipEvent_t eb)
{
  double* r  = reinterpret_cast<double*>(arena + L.r);
  double* p  = reinterpret_cast<double*>(arena + L.p);
  double* Ap = reinterpret_cast<double*>(arena + L.Ap);
  auto body = [&]() {
    launch_waxpby(m->nr, 1.0, r, 0.5, p, p, nullptr);
    launch_spmv(m, p, Ap, nullptr, nullptr);
    launch_waxpby(m->nr, 1.0, r, -1e-3, Ap, r, nullptr);
  };
  body(), body();
  float best = 1e30f;
  for (int rep = 0; rep < 2; rep++) {
    HIP_CHECK(hipEventRecord(ea, g.stream));
    for (int i = 0; i < 4; i++) body();
    HIP_CHECK(hipEventRecord(eb, g.stream));
    HIP_CHECK(hipEventSynchronize(eb));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
    best = std::min(best, 1e3f * ms / 4.f);
  }
  return best;
}
static void tune_matrix_placement(sb_matrix* m)
{
  const char* off = getenv("SB_PLACE");
  if ((off && atoi(off) == 0) || !g_tunePlacement) return;
  if (getenv("SB_SHARED_GPU") && atoi(getenv("SB_SHARED_GPU")) != 0) return; // ranks share this GPU (a rehearsal): timings mean nothing, memory is shared
  if (!m || !m->colInd || !m->val || m->nr == 0) return;
  if (m->fmt == 1 && m->C != 64) return;
  const size_t ne = place_elems(m), colBytes = ne * sizeof(uint32_t), valBytes = ne * sizeof(double);
  if (colBytes + valBytes < ((size_t)64 << 20)) return;
  const char* te   = getenv("SB_PLACE_TRIES");
  const int tries  = std::max(0, te ? atoi(te) : 4);
  const char* ve   = getenv("SB_PLACE_VEC_TRIES");
  const int vtries = std::max(1, ve ? atoi(ve) : 10);
  const size_t spacer = (size_t)(getenv("SB_PLACE_SPACER_MB") ? atol(getenv("SB_PLACE_SPACER_MB")) : 700) << 20;
  const char* re   = getenv("SB_PLACE_ROUNDS");
  const int rounds = std::max(1, re ? atoi(re) : 6);
  const int verbose = getenv("SB_PLACE_REPORT") ? atoi(getenv("SB_PLACE_REPORT")) : 0;
  const size_t colRegion = ((colBytes + ((size_t)2 << 20) - 1) >> 21) << 21, slabBytes = colRegion + valBytes + ((size_t)2 << 20);
  const VecLayout L = vec_layout(m->nr, m->nc, true, 0);
  HIP_CHECK(hipStreamSynchronize(g.stream));
  hipEvent_t ea, eb;
  HIP_CHECK(hipEventCreate(&ea));
  HIP_CHECK(hipEventCreate(&eb));
  const int mode = m->usePacked;
  m->usePacked   = 0; // the kernel that streams these arrays
  uint32_t* const col0 = m->colInd;
  double* const val0   = m->val;
  std::vector<char*> arenas, slabs, ballast; // everything tried stays allocated to the end: the next try lands on OTHER memory
  char *arena = nullptr, *home = nullptr; // the pair kept so far (home == nullptr: the stream where hipMalloc put it)
  float tFirst = 0.f, tBest = 1e30f, tWorst = 0.f;
  int timed = 0;
  size_t held = 0; // bytes the search holds at the moment
  auto room = [&](size_t bytes) { // (never more than a quarter of the device, nor than what is free with 2 GiB to spare)
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (held + bytes > totalB / 4) return false;
    if (bytes + ((size_t)2 << 30) > freeB) return false;
    held += bytes;
    return true;
  };
  auto point_stream = [&](char* sl) {
    m->colInd = sl ? reinterpret_cast<uint32_t*>(sl) : col0, m->val = sl ? reinterpret_cast<double*>(sl + colRegion) : val0;
  };
  int idle = 0; // consecutive rounds that found nothing better
  // no early exit for a device that looks flat: one process in four on devices that DO have fast pairs sees its first 52 pairs
  // within 5 % and finds a fast one among the next 50 (profiles/r04_placement_tuner_runs.txt)
  const bool managedFirst = !(getenv("SB_PLACE_MANAGED") && atoi(getenv("SB_PLACE_MANAGED")) == 0);
  const float flat = getenv("SB_PLACE_FLAT") ? (float)atof(getenv("SB_PLACE_FLAT")) : 0.f;
  // the search is over once the pair kept is >= 14.5 % faster than the slowest pair seen, or moves the proxy step's algorithmic
  // bytes at 6.0 TB/s (the fast level everywhere it has been seen: 131-133 us at HPCG 128^3 = 6.1 TB/s; the next level is 5.9)
  const float tFastEnough = (float)(1e6 * (sb_matrix_spmv_bytes(m) + 48.0 * m->nr) / 6.0e12);
  auto found = [&]() { return tBest <= 0.855f * tWorst || tBest <= tFastEnough; };
  for (int round = 0; round < rounds; round++) {
    const float tBefore = tBest;
    // the vectors' arena, stream in the home it has now
    for (int k = 0; k < vtries && room(L.total); k++) {
      char* q = nullptr;
      if (hipMalloc(&q, L.total) != hipSuccess) { (void)hipGetLastError(); break; }
      arenas.push_back(q);
      if (spacer && room(spacer)) { // the next arena lands `spacer` further on: a pair's speed follows how far apart its two
        char* sp = nullptr;         // allocations lie (within ~2.5 GB: the slow kind, profiles/r04_placement_lab7.txt)
        if (hipMalloc(&sp, spacer) == hipSuccess) ballast.push_back(sp);
        else (void)hipGetLastError();
      }
      HIP_CHECK(hipMemsetAsync(q, 0, L.total, g.stream));
      // r and p as a right-hand side would fill them (all-zero vectors would let the clock rise): 0x3f3f... = 4.8e-4
      HIP_CHECK(hipMemsetAsync(q + L.r, 0x3f, (size_t)m->nr * sizeof(double), g.stream));
      HIP_CHECK(hipMemsetAsync(q + L.p, 0x3f, (size_t)m->nr * sizeof(double), g.stream));
      const float t = placement_probe(m, q, L, ea, eb);
      if (timed++ == 0) tFirst = t; // what a process gets without looking: first allocation, stream where hipMalloc put it
      if (verbose > 1) fprintf(stderr, "sbhip placement: round %d arena %d (%p): %.2f us\n", round, k, (void*)q, t);
      tWorst = std::max(tWorst, t);
      if (t < tBest * (arena ? 0.985f : 1.0f)) tBest = t, arena = q; // (a new home has to be worth it: 1.5 %)
      if (found()) break;
    }
    if (!arena || found()) break;
    // the stream in fresh slabs
    for (int k = (round == 0 && managedFirst ? -1 : 0); k < tries && room(slabBytes); k++) {
      char* sl = nullptr;
      if (k < 0) {
        // the first candidate of all: MANAGED memory whose preferred location is this device, prefetched to it.  With the vectors
        // in plain hipMalloc memory such a stream ran at the fast level with every one of 8 arenas in 6 of 6 processes on devices
        // that have one (profiles/r04_placement_lab12.txt: 131-134 us where plain, uncached, fine-grained and contiguous slabs
        // allocated at the same moment gave 145-155); the vectors in managed memory are of the slow kind with any stream.  Where
        // managed memory is not device memory (no HMM), or on a device without fast pairs, the probe says so and it is dropped.
        if (hipMallocManaged((void**)&sl, slabBytes, hipMemAttachGlobal) != hipSuccess) { (void)hipGetLastError(); held -= slabBytes; continue; }
        if (hipMemAdvise(sl, slabBytes, hipMemAdviseSetPreferredLocation, g.device) != hipSuccess
            || hipMemPrefetchAsync(sl, slabBytes, g.device, g.stream) != hipSuccess
            || hipStreamSynchronize(g.stream) != hipSuccess) {
          (void)hipGetLastError();
          (void)hipFree(sl);
          held -= slabBytes;
          continue;
        }
      } else if (hipMalloc(&sl, slabBytes) != hipSuccess) { (void)hipGetLastError(); break; }
      slabs.push_back(sl);
      HIP_CHECK(hipMemcpy(sl, col0, colBytes, hipMemcpyDeviceToDevice));
      HIP_CHECK(hipMemcpy(sl + colRegion, val0, valBytes, hipMemcpyDeviceToDevice));
      point_stream(sl);
      // a new home for the stream is judged with several arenas, not only the best so far: which arena suits a stream depends
      // on where BOTH lie (profiles/r04_placement_lab7.txt: the level is a table over the two allocations' regions)
      char* const before = arena;
      char* partners[4] = { before, arenas.back(), arenas[arenas.size() / 2], arenas[arenas.size() / 4] };
      for (int a = 0; a < 4; a++) {
        bool dup = false;
        for (int b2 = 0; b2 < a; b2++) dup = dup || partners[b2] == partners[a];
        if (dup) continue;
        const float t = placement_probe(m, partners[a], L, ea, eb);
        timed++;
        tWorst = std::max(tWorst, t);
        if (verbose > 1) fprintf(stderr, "sbhip placement: round %d slab %d (%p) partner %d (%p): %.2f us\n", round, k, (void*)sl, a, (void*)partners[a], t);
        if (t < tBest * 0.985f) tBest = t, home = sl, arena = partners[a];
      }
      point_stream(home);
      if (found()) break; // (no need for the remaining slabs)
    }
    if (found()) break; // the pair kept is at the fast end (levels at HPCG 128^3: 131 | 137-139 | 143 | 147 | 153-157 us)
    idle = (round > 0 && tBest > tBefore * 0.985f) ? idle + 1 : 0;
    if (idle >= 3) break; // three rounds in a row without anything better: this device has nothing faster to offer
    if (flat > 0.f && round >= 1 && tWorst <= flat * tBest) break; // (SB_PLACE_FLAT=1.06: give up after two rounds within 6 %)
  }
  point_stream(home);
  HIP_CHECK(hipStreamSynchronize(g.stream));
  for (char* q : arenas)
    if (q != arena) HIP_CHECK(hipFree(q));
  for (char* sl : slabs)
    if (sl != home) HIP_CHECK(hipFree(sl));
  for (char* q : ballast) HIP_CHECK(hipFree(q));
  if (home) {
    m->slab = home, m->slabBytes = slabBytes, m->placeColMB = 0, m->placeValMB = 0;
    HIP_CHECK(hipFree(col0));
    HIP_CHECK(hipFree(val0));
  }
  m->vecArena = arena, m->vecArenaBytes = arena ? L.total : 0, m->vecArenaBusy = false;
  m->placeTried = timed;
  m->placeUs[0] = tFirst, m->placeUs[1] = tBest, m->placeUs[2] = tWorst;
  m->usePacked  = mode;
  HIP_CHECK(hipEventDestroy(ea));
  HIP_CHECK(hipEventDestroy(eb));
  if (getenv("SB_PLACE_REPORT"))
    fprintf(stderr, "sbhip placement: stream of %.1f MB + vectors' arena of %.1f MB: proxy step %.2f us with the first arena tried and "
        "the stream where hipMalloc put it, %.2f us at the pair kept (%s; %d probes, slowest %.2f us)\n",
        1e-6 * (double)(colBytes + valBytes), 1e-6 * (double)L.total, tFirst, tBest, home ? "stream moved to a fresh slab" : "stream stays",
        timed, tWorst);
}

// ---- lab calls (tools/placement_lab6.py; uploads made with SB_PLACE=0) --------------------------------------------------------
// the stream copied to memory the CALLER provides (col: place_elems x 4 bytes, val: x 8 bytes; nothing is freed or owned here)
void sb_matrix_place_at(sb_matrix* m, void* colMem, void* valMem)
{
  need_init();
  if (m->placeTried) SB_FATAL("sb_matrix_place_at is a lab call: upload with SB_PLACE=0");
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (!m->colInd0) m->colInd0 = m->colInd, m->val0 = m->val;
  const size_t ne = place_elems(m);
  HIP_CHECK(hipMemcpy(colMem, m->colInd0, ne * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  HIP_CHECK(hipMemcpy(valMem, m->val0, ne * sizeof(double), hipMemcpyDeviceToDevice));
  m->colInd = (uint32_t*)colMem, m->val = (double*)valMem;
}
// back to the first upload (so that sb_matrix_free finds what it allocated)
void sb_matrix_place_home(sb_matrix* m)
{
  if (m->colInd0) m->colInd = m->colInd0, m->val = m->val0, m->colInd0 = nullptr, m->val0 = nullptr;
}
// the tuner's proxy step (us) with the loop's vectors laid out at `arena` (sb_placement_arena_bytes of caller's memory)
size_t sb_placement_arena_bytes(const sb_matrix* m) { return vec_layout(m->nr, m->nc, true, 0).total; }
float sb_placement_probe(sb_matrix* m, void* arena)
{
  need_init();
  const VecLayout L = vec_layout(m->nr, m->nc, true, 0);
  HIP_CHECK(hipMemsetAsync(arena, 0, L.total, g.stream));
  HIP_CHECK(hipMemsetAsync((char*)arena + L.r, 0x3f, (size_t)m->nr * sizeof(double), g.stream));
  HIP_CHECK(hipMemsetAsync((char*)arena + L.p, 0x3f, (size_t)m->nr * sizeof(double), g.stream));
  hipEvent_t ea, eb;
  HIP_CHECK(hipEventCreate(&ea));
  HIP_CHECK(hipEventCreate(&eb));
  const int mode = m->usePacked;
  m->usePacked   = 0;
  const float t  = placement_probe(m, (char*)arena, L, ea, eb);
  m->usePacked   = mode;
  HIP_CHECK(hipEventDestroy(ea));
  HIP_CHECK(hipEventDestroy(eb));
  return t;
}
