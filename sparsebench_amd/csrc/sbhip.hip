// sbhip.hip -- implementation of the C-ABI in include/sbhip.h.
//
// One process drives one MI355X.  All work goes to one HIP stream; the CG loop
// keeps alpha, beta, rtrans and the loop-exit flag in HBM so no host round trip
// happens between iterations.  RCCL is opened lazily (dlopen) only when a
// communicator is attached, so single-GPU runs never load it.
#include "../../include/sbhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kernels.hip.h"
#include "pack.hip.h"

#include <algorithm>
#include <iterator>
#include <string>
#include <unordered_map>

using namespace sbk;

#define HIP_CHECK(call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "sbhip: HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, \
          __LINE__, hipGetErrorString(e_));                                               \
      exit(EXIT_FAILURE);                                                                 \
    }                                                                                     \
  } while (0)

#define SB_FATAL(...)                                        \
  do {                                                       \
    fprintf(stderr, "sbhip: %s:%d: ", __FILE__, __LINE__);   \
    fprintf(stderr, __VA_ARGS__);                            \
    fprintf(stderr, "\n");                                   \
    exit(EXIT_FAILURE);                                      \
  } while (0)

// ---------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess_ = 0 };
enum { ncclInt8_ = 0, ncclInt32_ = 2, ncclFloat64_ = 8 }; // ncclDataType_t
enum { ncclSum_ = 0, ncclMax_ = 2 }; // ncclRedOp_t

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId*);
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  int (*CommDestroy)(ncclComm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  const char* (*GetErrorString)(int);
} rccl;

void rccl_open()
{
  if (rccl.h) return;
  const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for (const char* n : names) {
    rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (rccl.h) break;
  }
  if (!rccl.h) SB_FATAL("cannot open librccl: %s", dlerror());
#define BIND(field, sym)                                           \
  *(void**)(&rccl.field) = dlsym(rccl.h, sym);                     \
  if (!rccl.field) SB_FATAL("librccl lacks %s", sym)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(AllReduce, "ncclAllReduce");
  BIND(AllGather, "ncclAllGather");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
}
#define RCCL_CHECK(call)                                                                  \
  do {                                                                                    \
    int r_ = (call);                                                                      \
    if (r_ != ncclSuccess_)                                                               \
      SB_FATAL("RCCL error %d (%s)", r_, rccl.GetErrorString ? rccl.GetErrorString(r_) : "?"); \
  } while (0)

struct Ctx {
  bool init = false;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr; // halo exchange while the interior tiles are multiplied
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  hipDeviceProp_t prop;
  char name[320];
  // scratch for stand-alone ddot / permuted sb_spmv
  double* partials = nullptr;
  size_t partialsCap = 0;
  double* scalar = nullptr; // one device double
  double* ws[2] = { nullptr, nullptr };
  size_t wsCap[2] = { 0, 0 };
  // communicator: RCCL, or a caller-provided host-mediated transport (sb_comm_init_transport)
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
  bool hasXport = false;
  sb_transport xport;
  // in-kernel all-reduce over peer-mapped memory (kernels.hip.h: p2p_allreduce_sum)
  P2PSlot* p2pBuf = nullptr;        // this rank's buffer (fine-grained device memory), P2PSlot[2][P2P_MAX]
  void* p2pPeer[P2P_MAX] = {};      // peers' buffers as opened here (IPC); [rank] = p2pBuf
  P2PView* p2pView = nullptr;       // device copy of the view
  unsigned long long p2pSeq = 0;    // exchanges issued so far (identical on every rank)
  bool p2pOn = false;
} g;

inline bool multi_rank() { return g.comm != nullptr || g.hasXport; }
// a stop flag that is never set, for launches outside a CG loop (kernels that fetch the flag
// together with other data want a valid address)
inline const int* zero_flag() { return reinterpret_cast<const int*>(g.scalar + 4); }

void need_init()
{
  if (!g.init) SB_FATAL("sb_init() has not been called");
}

double* scratch_partials(size_t n)
{
  if (n > g.partialsCap) {
    if (g.partials) HIP_CHECK(hipFree(g.partials));
    g.partialsCap = n + n / 2 + 1024;
    HIP_CHECK(hipMalloc(&g.partials, g.partialsCap * sizeof(double)));
  }
  return g.partials;
}

double* scratch_ws(int which, size_t n)
{
  if (n > g.wsCap[which]) {
    if (g.ws[which]) HIP_CHECK(hipFree(g.ws[which]));
    g.wsCap[which] = n;
    HIP_CHECK(hipMalloc(&g.ws[which], n * sizeof(double)));
  }
  return g.ws[which];
}

inline uint32_t stream_grid(uint32_t nWork, uint32_t perBlock)
{ // memory-bound grid: enough blocks to fill 256 CUs x 8, grid-stride the rest
  uint32_t b   = (nWork + perBlock - 1) / perBlock;
  uint32_t cap = (uint32_t)g.prop.multiProcessorCount * 8u;
  if (b > cap) b = cap;
  return b ? b : 1;
}
} // namespace

struct sb_matrix {
  int fmt; // 0 CRS, 1 SCS
  uint32_t nr, nc, nnz;
  // CRS
  uint32_t *rowPtr = nullptr, *rowBlocks = nullptr;
  uint32_t nRowBlocks = 0;
  // SCS
  uint32_t C = 0, sigma = 0, nChunks = 0, nElems = 0, nrPadded = 0;
  uint32_t *chunkPtr = nullptr, *chunkLens = nullptr;
  uint32_t *oldToNew = nullptr, *newToOld = nullptr; // device, nr each (SCS permuted only)
  int permuted = 0;
  // both
  uint32_t* colInd = nullptr;
  double* val = nullptr;
  // SCS C=64: device-private compressed mirror (pack.hip.h)
  PackMeta* pmeta = nullptr;
  uint32_t *pidx = nullptr, *pcodes = nullptr;
  double* pdict = nullptr;
  int packLevel = 0; // 0 none, 1 16-bit columns, 2 + value dictionary
  int usePacked = 0;
  uint32_t padCol = 0;
  double packedBytes = 0.0; // matrix-stream bytes of the packed form
  uint32_t nWideChunks = 0;
  int nDict = 0;
  // level 3: per-tile x windows staged in LDS
  uint32_t* tileSegPtr = nullptr;
  TileSeg* tileSegs    = nullptr;
  uint32_t* pslots     = nullptr;
  uint32_t ldsWindow   = 0; // doubles, incl. slot 0
  double slotBytes     = 0.0;
  // level 4: pattern dictionary (one byte per element)
  uint16_t* rowBase     = nullptr;
  uint32_t *tileClass = nullptr, *jcodes = nullptr;
  PatEntry* classDict   = nullptr;
  TileHdr* tileHdrs     = nullptr;
  PatEntry* rowPats     = nullptr; // level 5: shared row patterns
  PatEntry* excRows     = nullptr; // level 5: expanded exception rows of the U chunks
  uint32_t patDict = 0, patExcLds = 0; // LDS layout of spmv_scs64_pat: table entries, exception entries
  uint32_t patInterior = 0;            // headers [0, patInterior): tiles that touch no halo column
  // CRS: a private Sell-64-1 mirror carrying only the pattern levels (SKIPPAD kernel); usePacked
  // 3 = product through the mirror, 0 = native CRS kernel
  sb_matrix* mirror = nullptr;
  uint32_t nRowPats = 0, nUniformChunks = 0;
  uint32_t nPatClasses  = 0;
  double patBytes       = 0.0;
};

struct sb_halo {
  uint32_t nr;
  int outdegree, indegree, totalSend, externalCount;
  std::vector<int> destinations, sendCounts, sdispls, sources, recvCounts, rdispls;
  uint32_t* packIdx = nullptr; // device: row (in the vector's order) of each sent element
  double* sendBuf = nullptr;   // device
};

struct sb_cg {
  const sb_matrix* A;
  sb_halo* halo;
  uint32_t nr, nc;
  double *r, *p, *Ap, *x, *b, *xexact;
  CgScalars* S;
  double* partials;
  uint32_t nPartials;
  double *rr_hist, *pAp_hist;
  int hist_cap;
  int fused, use_graph;
  hipGraphExec_t iterGraph;
  bool graphReady;
  double region_ms[4];
  std::vector<hipEvent_t> ev;
  std::vector<int> evRegion;
  size_t evUsed;
  bool timing;
  hipEvent_t evLoop0, evLoop1;
  float loop_ms;
  // optional in-situ timing of every SpMV launch (bench.py's roofline leg)
  bool spmvTiming;
  std::vector<hipEvent_t> spmvEv;
  size_t spmvEvUsed;
  int k_next;        // next loop body to enqueue
  bool started;
  CgScalars hostS;   // staging copy for the H2D of the control block
};

// ===========================================================================
// context
// ===========================================================================
// Every sb_* function below is declared extern "C" by include/sbhip.h, which fixes
// its linkage; the helpers in between stay C++.

const char* sb_version(void) { return "sparsebench_amd sbhip 0.1 (gfx950)"; }

int sb_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void sb_init(int device)
{
  if (g.init) {
    if (device != g.device) SB_FATAL("already initialised on device %d", g.device);
    return;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    SB_FATAL("no HIP device visible (%s): the HIP path is the only path", hipGetErrorString(e));
  if (device < 0 || device >= n) SB_FATAL("device %d out of range (%d visible)", device, n);
  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipGetDeviceProperties(&g.prop, device));
  snprintf(g.name, sizeof g.name, "%s (%s)", g.prop.name[0] ? g.prop.name : "AMD Instinct", g.prop.gcnArchName);
  HIP_CHECK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&g.evFork, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&g.evJoin, hipEventDisableTiming));
  HIP_CHECK(hipMalloc(&g.scalar, 64));
  HIP_CHECK(hipMemset(g.scalar, 0, 64)); // [0] scratch double, [4] a permanent int 0 (zero_flag)
  g.device = device;
  g.init   = true;
}

void sb_finalize(void)
{
  if (!g.init) return;
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (g.comm) sb_comm_finalize();
  if (g.partials) HIP_CHECK(hipFree(g.partials));
  for (int i = 0; i < 2; i++)
    if (g.ws[i]) HIP_CHECK(hipFree(g.ws[i]));
  HIP_CHECK(hipFree(g.scalar));
  HIP_CHECK(hipStreamDestroy(g.stream));
  HIP_CHECK(hipStreamDestroy(g.stream2));
  HIP_CHECK(hipEventDestroy(g.evFork));
  HIP_CHECK(hipEventDestroy(g.evJoin));
  g = Ctx();
}

int sb_is_initialized(void) { return g.init ? 1 : 0; }
const char* sb_device_name(void) { need_init(); return g.name; }
int sb_num_cus(void) { need_init(); return g.prop.multiProcessorCount; }
void sb_sync(void) { need_init(); HIP_CHECK(hipStreamSynchronize(g.stream)); }
void* sb_stream(void) { need_init(); return (void*)g.stream; }

void* sb_malloc(size_t bytes)
{
  need_init();
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
  return p;
}
void sb_free(void* dev)
{
  if (dev) HIP_CHECK(hipFree(dev));
}
void sb_memset(void* dev, int byte, size_t bytes)
{
  need_init();
  HIP_CHECK(hipMemsetAsync(dev, byte, bytes, g.stream));
}
void sb_h2d(void* dev, const void* host, size_t bytes)
{
  need_init();
  HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void sb_d2h(void* host, const void* dev, size_t bytes)
{
  need_init();
  HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void sb_d2d(void* dst, const void* src, size_t bytes)
{
  need_init();
  HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g.stream));
}
int sb_is_device_ptr(const void* p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return a.type == hipMemoryTypeDevice ? 1 : 0;
}

void* sb_event_create(void)
{
  need_init();
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return (void*)e;
}
void sb_event_record(void* ev) { HIP_CHECK(hipEventRecord((hipEvent_t)ev, g.stream)); }
float sb_event_elapsed_ms(void* a, void* b)
{
  float ms = 0.f;
  HIP_CHECK(hipEventSynchronize((hipEvent_t)b));
  HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
  return ms;
}
void sb_event_destroy(void* ev) { HIP_CHECK(hipEventDestroy((hipEvent_t)ev)); }

// ===========================================================================
// matrices
// ===========================================================================
static void* upload(const void* host, size_t bytes)
{
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, bytes ? bytes : 8));
  if (bytes) HIP_CHECK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  return d;
}

static void build_crs_mirror(sb_matrix* m, const uint32_t* rowPtr, const uint32_t* colInd, const double* val);

sb_matrix* sb_crs_upload(uint32_t nr, uint32_t nc, const uint32_t* rowPtr, const uint32_t* colInd,
    const double* val)
{
  need_init();
  sb_matrix* m = new sb_matrix();
  m->fmt = 0, m->nr = nr, m->nc = nc, m->nnz = rowPtr[nr];
  for (uint32_t i = 0; i < nr; i++)
    if (rowPtr[i + 1] < rowPtr[i]) SB_FATAL("CRS rowPtr not monotone at row %u", i);
  for (uint32_t k = 0; k < m->nnz; k++)
    if (colInd[k] >= nc) SB_FATAL("CRS colInd[%u]=%u out of range (nc=%u)", k, colInd[k], nc);
  // Row blocks: as many rows as fit CRS_TILE nonzeros and CRS_THREADS rows; a row
  // longer than the tile gets a block of its own.
  std::vector<uint32_t> rb;
  rb.push_back(0);
  uint32_t r = 0;
  while (r < nr) {
    uint32_t start = r, base = rowPtr[r];
    while (r < nr && r - start < (uint32_t)CRS_THREADS && rowPtr[r + 1] - base <= (uint32_t)CRS_TILE) r++;
    if (r == start) r++; // single oversize row
    rb.push_back(r);
  }
  m->nRowBlocks = (uint32_t)rb.size() - 1;
  m->rowBlocks  = (uint32_t*)upload(rb.data(), rb.size() * sizeof(uint32_t));
  m->rowPtr     = (uint32_t*)upload(rowPtr, ((size_t)nr + 1) * sizeof(uint32_t));
  m->colInd     = (uint32_t*)upload(colInd, (size_t)m->nnz * sizeof(uint32_t));
  m->val        = (double*)upload(val, (size_t)m->nnz * sizeof(double));
  build_crs_mirror(m, rowPtr, colInd, val);
  return m;
}

__global__ void remap_cols_k(uint32_t n, uint32_t nr, const uint32_t* __restrict__ oldToNew,
    uint32_t* colInd)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t c = colInd[i];
    if (c < nr) colInd[i] = oldToNew[c];
  }
}

// Build the compressed mirror of an uploaded SCS C=64 matrix (pack.hip.h).  hostVal is
// the host copy of val (dictionary detection happens on the host, with early exit).
static void build_packed(sb_matrix* m, const double* hostVal, const uint32_t* oldToNewPerm)
{
  if (m->fmt != 1 || m->C != 64 || m->nChunks == 0) return;
  const char* env = getenv("SB_PACK");
  const int want  = env ? atoi(env) : 2; // 0 off, 1 columns only, 2 columns + values
  if (want <= 0) return;
  // 1. value dictionary (<= 256 distinct bit patterns, +0.0 always present for padding)
  std::vector<unsigned long long> dict;
  dict.push_back(0ull);
  bool dictOk = want >= 2;
  if (dictOk) {
    unsigned long long last = 0ull;
    for (size_t i = 0; i < m->nElems; i++) {
      unsigned long long b;
      memcpy(&b, hostVal + i, 8);
      if (b == last) continue;
      last = b;
      if (std::find(dict.begin(), dict.end(), b) == dict.end()) {
        dict.push_back(b);
        if (dict.size() > 256) {
          dictOk = false;
          break;
        }
      }
    }
  }
  std::sort(dict.begin(), dict.end());
  m->nDict  = dictOk ? (int)dict.size() : 0;
  m->padCol = (m->permuted && oldToNewPerm && m->nr) ? oldToNewPerm[0] : 0u;
  // 2. per-chunk column range on the device
  uint32_t *cmin = nullptr, *cmax = nullptr;
  HIP_CHECK(hipMalloc(&cmin, (size_t)m->nChunks * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&cmax, (size_t)m->nChunks * sizeof(uint32_t)));
  const dim3 grid((m->nChunks + 3) / 4), block(256);
  hipLaunchKernelGGL(pack_minmax_k, grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd, m->val,
      m->nChunks, m->padCol, cmin, cmax);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> lo(m->nChunks), hi(m->nChunks), lens(m->nChunks);
  sb_d2h(lo.data(), cmin, lo.size() * sizeof(uint32_t));
  sb_d2h(hi.data(), cmax, hi.size() * sizeof(uint32_t));
  sb_d2h(lens.data(), m->chunkLens, lens.size() * sizeof(uint32_t));
  HIP_CHECK(hipFree(cmin));
  HIP_CHECK(hipFree(cmax));
  // 3. stream positions
  std::vector<PackMeta> meta(m->nChunks);
  uint64_t grp = 0, units = 0;
  m->nWideChunks = 0;
  for (uint32_t c = 0; c < m->nChunks; c++) {
    const bool empty = lo[c] > hi[c];
    const bool wide  = !empty && (hi[c] - lo[c]) >= 0xFFFFu;
    const uint32_t ng = (lens[c] + 3u) / 4u;
    meta[c].grp    = (uint32_t)grp;
    meta[c].idxOff = (uint32_t)units;
    meta[c].base   = empty ? 0u : lo[c];
    meta[c].info   = lens[c] | (wide ? 0x80000000u : 0u);
    grp += ng;
    units += (uint64_t)ng * (wide ? 2u : 1u);
    m->nWideChunks += wide;
  }
  if (grp > 0xFFFFFFFFull || units > 0xFFFFFFFFull) return; // does not fit the 32-bit positions
  m->pmeta = (PackMeta*)upload(meta.data(), meta.size() * sizeof(PackMeta));
  HIP_CHECK(hipMalloc(&m->pidx, (size_t)units * 512 + 1024));
  unsigned long long* dbits = nullptr;
  if (m->nDict) {
    HIP_CHECK(hipMalloc(&m->pcodes, (size_t)grp * 256 + 1024));
    std::vector<unsigned long long> padded(256, 0ull);
    std::copy(dict.begin(), dict.end(), padded.begin());
    dbits    = (unsigned long long*)upload(dict.data(), dict.size() * sizeof(unsigned long long));
    m->pdict = (double*)upload(padded.data(), 256 * sizeof(double));
  }
  hipLaunchKernelGGL(pack_write_k, grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd, m->val,
      m->pmeta, dbits, m->nDict, m->nChunks, m->padCol, m->pidx, m->pcodes);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (dbits) HIP_CHECK(hipFree(dbits));
  m->packLevel   = m->nDict ? 2 : 1;
  m->usePacked   = 1;
  m->packedBytes = (double)units * 512.0 + (m->nDict ? (double)grp * 256.0 : 8.0 * m->nElems) +
                   16.0 * m->nChunks;
}

// Level 3 of the compressed mirror: per tile (4 chunks = one workgroup) the contiguous
// column ranges its rows touch, so the kernel can stage them in LDS (pack.hip.h).
// Host arrays are the reference-layout ones (columns in ORIGINAL numbering).
static void build_lds_windows(sb_matrix* m, const uint32_t* chunkPtr, const uint32_t* chunkLens,
    const uint32_t* colInd, const double* val, const uint32_t* oldToNewPerm)
{
  if (m->packLevel < 1) return;
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 3) < 3) return;
  const uint32_t WMAX = 6144, MERGE_GAP = 8; // window <= 48 KiB of LDS per workgroup
  const uint32_t nTiles = (m->nChunks + 3) / 4;
  std::vector<uint32_t> segPtr(nTiles + 1, 0);
  std::vector<TileSeg> segs;
  std::vector<uint32_t> cols;
  std::vector<uint64_t> bitmap;
  uint32_t maxWin = 0;
  uint64_t sumWin = 0, sumElems = 0;
  for (uint32_t t = 0; t < nTiles; t++) {
    cols.clear();
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint32_t c = t * 4; c < std::min(t * 4 + 4, m->nChunks); c++) {
      const size_t cp = chunkPtr[c];
      const size_t n  = (size_t)chunkLens[c] * 64;
      for (size_t e = 0; e < n; e++) {
        uint32_t col = colInd[cp + e];
        unsigned long long bits;
        memcpy(&bits, val + cp + e, 8);
        if (col == 0 && bits == 0) continue; // padding (or an explicit 0.0 at column 0): slot 0
        if (m->permuted && col < m->nr) col = oldToNewPerm[col];
        cols.push_back(col);
        lo = std::min(lo, col), hi = std::max(hi, col);
      }
    }
    segPtr[t] = (uint32_t)segs.size();
    if (cols.empty()) continue;
    // distinct columns in ascending order: bitmap when the span is modest, sort otherwise
    const uint64_t span = (uint64_t)hi - lo + 1;
    uint32_t win = 1; // slot 0
    auto emit = [&](uint32_t first, uint32_t last) {
      TileSeg s;
      s.col = first, s.len = last - first + 1, s.lds = win, s.pad_ = 0;
      win += s.len;
      segs.push_back(s);
    };
    if (span <= (1u << 22)) {
      bitmap.assign((span + 63) / 64, 0ull);
      for (uint32_t c : cols) bitmap[(c - lo) >> 6] |= 1ull << ((c - lo) & 63);
      bool open = false;
      uint32_t first = 0, last = 0;
      for (uint64_t w = 0; w < bitmap.size(); w++) {
        uint64_t bits = bitmap[w];
        while (bits) {
          const uint32_t c = lo + (uint32_t)(w * 64 + (uint64_t)__builtin_ctzll(bits));
          bits &= bits - 1;
          if (!open) first = last = c, open = true;
          else if (c - last <= MERGE_GAP) last = c;
          else emit(first, last), first = last = c;
        }
      }
      if (open) emit(first, last);
    } else {
      std::sort(cols.begin(), cols.end());
      uint32_t first = cols[0], last = cols[0];
      for (uint32_t c : cols) {
        if (c - last <= MERGE_GAP) last = std::max(last, c);
        else emit(first, last), first = last = c;
      }
      emit(first, last);
    }
    if (win > WMAX) return; // some tile's window does not fit LDS: stay at level 1/2
    maxWin = std::max(maxWin, win);
    sumWin += win;
    sumElems += cols.size();
  }
  segPtr[nTiles] = (uint32_t)segs.size();
  // Staging pays only when a window entry is reused several times and the window is made
  // of long runs (coalesced copies).  Measured: 27-pt stencil reuse 4.6 / run ~510 ->
  // 1.15x faster than gathering through the cache; irregular FE-like matrix with 5 % far
  // couplings reuse 2.5 / run ~3 -> 2.5x slower.  SB_PACK_LDS=1 forces it on, =0 off.
  {
    const double reuse = sumWin ? (double)sumElems / (double)sumWin : 0.0;
    const double run   = segs.empty() ? 0.0 : (double)sumWin / (double)segs.size();
    const char* force  = getenv("SB_PACK_LDS");
    const bool want    = force ? atoi(force) != 0 : (reuse >= 3.0 && run >= 32.0);
    if (!want) return;
  }
  if (maxWin == 0) maxWin = 1;
  TileSeg dummy = { 0, 0, 0, 0 };
  if (segs.empty()) segs.push_back(dummy);
  m->tileSegPtr = (uint32_t*)upload(segPtr.data(), segPtr.size() * sizeof(uint32_t));
  m->tileSegs   = (TileSeg*)upload(segs.data(), segs.size() * sizeof(TileSeg));
  std::vector<PackMeta> meta(m->nChunks);
  sb_d2h(meta.data(), m->pmeta, meta.size() * sizeof(PackMeta));
  const uint64_t groups = meta.empty() ? 0 : (uint64_t)meta.back().grp + ((meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
  HIP_CHECK(hipMalloc(&m->pslots, (size_t)groups * 512 + 1024));
  hipLaunchKernelGGL(pack_slots_k, dim3(nTiles), dim3(256), 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd,
      m->val, m->pmeta, m->tileSegPtr, m->tileSegs, m->nChunks, m->padCol, m->pslots);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  m->ldsWindow = maxWin;
  m->slotBytes = (double)groups * 512.0 + (m->nDict ? (double)groups * 256.0 : 8.0 * m->nElems) +
                 16.0 * m->nChunks + 16.0 * segs.size() + 4.0 * nTiles;
  m->usePacked = 2; // 2: packed stream + x window in LDS
}

// Level 4: one byte per element naming a (value, slot delta) pair of the tile's class;
// level 5: per chunk one shared row pattern + the odd lanes (pack.hip.h).  Needs the value
// dictionary and the LDS windows.  SB_PACK=4 stops at level 4 (every chunk per-lane).
static void build_patterns(sb_matrix* m)
{
  if (m->usePacked != 2 || m->nDict <= 0) return;
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 4) < 4) return;
  const uint32_t nTiles = (m->nChunks + 3) / 4;
  uint32_t *dCount = nullptr, *dKeys = nullptr;
  HIP_CHECK(hipMalloc(&m->rowBase, (size_t)m->nChunks * 64 * sizeof(uint16_t) + 16));
  HIP_CHECK(hipMalloc(&dCount, (size_t)nTiles * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&dKeys, (size_t)nTiles * 256 * sizeof(uint32_t)));
  hipLaunchKernelGGL(pat_collect_k, dim3(nTiles), dim3(256), 0, g.stream, m->pmeta, m->pslots, m->pcodes,
      m->nChunks, m->rowBase, dCount, dKeys);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> count(nTiles), keys((size_t)nTiles * 256);
  sb_d2h(count.data(), dCount, count.size() * sizeof(uint32_t));
  sb_d2h(keys.data(), dKeys, keys.size() * sizeof(uint32_t));
  HIP_CHECK(hipFree(dCount));
  HIP_CHECK(hipFree(dKeys));
  auto giveUp = [&]() { sb_free(m->rowBase), m->rowBase = nullptr; };
  for (uint32_t t = 0; t < nTiles; t++)
    if (count[t] > PAT_MAX) return giveUp();
  // tiles -> classes of <= PAT_MAX pairs: a class that already holds the tile's pairs,
  // else the first class the pairs still fit into, else a new class
  std::vector<std::vector<uint32_t>> classes;
  std::vector<uint32_t> tileClass(nTiles, 0), merged;
  const size_t maxClasses = std::max<size_t>(64, nTiles / 4);
  uint32_t lastClass = 0;
  for (uint32_t t = 0; t < nTiles; t++) {
    uint32_t* k = keys.data() + (size_t)t * 256;
    std::sort(k, k + count[t]);
    int found = -1;
    if (!classes.empty() && std::includes(classes[lastClass].begin(), classes[lastClass].end(), k, k + count[t]))
      found = (int)lastClass;
    for (size_t c = 0; found < 0 && c < classes.size(); c++)
      if (std::includes(classes[c].begin(), classes[c].end(), k, k + count[t])) found = (int)c;
    for (size_t c = 0; found < 0 && c < classes.size(); c++) {
      merged.clear();
      std::set_union(classes[c].begin(), classes[c].end(), k, k + count[t], std::back_inserter(merged));
      if (merged.size() <= PAT_MAX) classes[c] = merged, found = (int)c;
    }
    if (found < 0) {
      if (classes.size() >= maxClasses) return giveUp(); // no repeating patterns: not worth the tables
      classes.emplace_back(k, k + count[t]);
      found = (int)classes.size() - 1;
    }
    tileClass[t] = lastClass = (uint32_t)found;
  }
  if (classes.empty()) classes.emplace_back();
  std::vector<double> dict(256);
  sb_d2h(dict.data(), m->pdict, 256 * sizeof(double));
  std::vector<uint32_t> classKeys(classes.size() * 256, PAT_EMPTY);
  std::vector<PatEntry> classDict(classes.size() * 256, PatEntry{ 0.0, 0, 0u });
  for (size_t c = 0; c < classes.size(); c++)
    for (size_t i = 0; i < classes[c].size(); i++) {
      const uint32_t key       = classes[c][i];
      classKeys[c * 256 + i]   = key;
      PatEntry& e              = classDict[c * 256 + i];
      e.v                      = dict[key & 255u];
      if (key & PAT_ABS) e.off8 = 0u, e.m = 0u; // padding: slot 0
      else e.off8 = (uint32_t)(8 * ((int32_t)((key >> 8) & 0xFFFFu) - 32768)), e.m = 1u; // 8 * (slot - rowBase), mod 2^32
    }
  uint32_t* dClassKeys = (uint32_t*)upload(classKeys.data(), classKeys.size() * sizeof(uint32_t));
  m->tileClass         = (uint32_t*)upload(tileClass.data(), tileClass.size() * sizeof(uint32_t));
  m->classDict         = (PatEntry*)upload(classDict.data(), classDict.size() * sizeof(PatEntry));
  std::vector<PackMeta> meta(m->nChunks);
  sb_d2h(meta.data(), m->pmeta, meta.size() * sizeof(PackMeta));
  const uint64_t groups = (uint64_t)meta.back().grp + ((meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
  for (const PackMeta& pm : meta)
    if ((pm.info & 0x7FFFFFFFu) >= PAT_NOPAD) { // chunk width collides with the header's flag bits
      sb_free(dClassKeys), sb_free(m->tileClass), sb_free(m->classDict);
      m->tileClass = nullptr, m->classDict = nullptr;
      return giveUp();
    }
  uint32_t* lanes = nullptr; // per-lane code words, group-major (the L form of every chunk)
  HIP_CHECK(hipMalloc(&lanes, (size_t)groups * 256 + 1024));
  hipLaunchKernelGGL(pat_encode_k, dim3(nTiles), dim3(256), 0, g.stream, m->pmeta, m->pslots, m->pcodes,
      m->nChunks, m->rowBase, m->tileClass, dClassKeys, lanes);
  HIP_CHECK(hipGetLastError());
  // Level 5 (row patterns): dominant code sequence and exception lanes of every chunk
  const bool wantRows = (env ? atoi(env) : 5) >= 5;
  uint32_t *dDom = nullptr, *dExc = nullptr;
  HIP_CHECK(hipMalloc(&dDom, (size_t)groups * sizeof(uint32_t) + 16));
  HIP_CHECK(hipMalloc(&dExc, (size_t)m->nChunks * 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(pat_dominant_k, dim3(nTiles), dim3(256), 0, g.stream, m->pmeta, lanes, m->nChunks, dDom,
      dExc);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> dom(groups ? groups : 1), exc((size_t)m->nChunks * 2);
  sb_d2h(dom.data(), dDom, (size_t)groups * sizeof(uint32_t));
  sb_d2h(exc.data(), dExc, exc.size() * sizeof(uint32_t));
  HIP_CHECK(hipFree(dDom));
  // chunk by chunk: U (row pattern + expanded exception lanes) or L (code words of all 64
  // lanes); row patterns are shared between chunks (key: the expanded entries).  A tile's
  // exception entries are staged in LDS, so a tile with too many of them stays L.
  const size_t maxPatEntries = 1u << 20; // 16 MiB of pattern rows at most
  std::vector<PatEntry> rowPats;
  std::unordered_map<std::string, uint32_t> patIndex;
  std::vector<uint32_t> chunkOff(m->nChunks), chunkFlags(m->nChunks), chunkPat(m->nChunks, 0);
  std::vector<uint32_t> tileExcStart(nTiles, 0), tileExcCount(nTiles, 0);
  std::vector<PatEntry> row;
  uint64_t words = 0, excEntries = 0;
  uint32_t excLds = 0;
  bool anyL       = false;
  m->nUniformChunks = 0;
  auto n_exc = [&](uint32_t c) {
    return (uint32_t)__builtin_popcount(exc[2 * (size_t)c]) + (uint32_t)__builtin_popcount(exc[2 * (size_t)c + 1]);
  };
  for (uint32_t t = 0; t < nTiles; t++) {
    const uint32_t c0 = t * 4, c1 = std::min(c0 + 4, m->nChunks);
    uint64_t tileExc = 0;
    bool tileOk      = wantRows;
    for (uint32_t c = c0; c < c1 && tileOk; c++) {
      const uint32_t len = meta[c].info & 0x7FFFFFFFu;
      if (len == 0 || n_exc(c) > PAT_EXC_MAX) continue; // this chunk will be L
      tileExc += (uint64_t)n_exc(c) * len;
    }
    if (tileExc > PAT_EXC_LDS_MAX) tileOk = false;
    tileExcStart[t] = (uint32_t)excEntries;
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t len = meta[c].info & 0x7FFFFFFFu, ng = (len + 3u) / 4u, nExc = n_exc(c);
      bool uni = tileOk && len > 0 && nExc <= PAT_EXC_MAX, nopad = true;
      if (uni) {
        row.resize(len);
        const PatEntry* cd = classDict.data() + (size_t)tileClass[t] * 256;
        for (uint32_t j = 0; j < len; j++) {
          row[j] = cd[(dom[meta[c].grp + j / 4] >> (8u * (j & 3u))) & 255u];
          nopad  = nopad && row[j].m == 1u;
        }
        std::string key((const char*)row.data(), row.size() * sizeof(PatEntry));
        auto it = patIndex.find(key);
        if (it != patIndex.end()) chunkPat[c] = it->second;
        else if (rowPats.size() + len <= maxPatEntries) {
          chunkPat[c] = (uint32_t)rowPats.size();
          patIndex.emplace(std::move(key), chunkPat[c]);
          rowPats.insert(rowPats.end(), row.begin(), row.end());
        } else uni = false; // table full
      }
      if (uni) {
        chunkOff[c]   = (uint32_t)excEntries;
        chunkFlags[c] = len | PAT_UNIFORM | (nopad ? PAT_NOPAD : 0u);
        excEntries += (uint64_t)nExc * len;
        tileExcCount[t] += nExc * len;
        m->nUniformChunks++;
      } else {
        chunkOff[c]   = (uint32_t)words;
        chunkFlags[c] = len;
        words += (uint64_t)ng * 64u;
        anyL = anyL || len > 0;
      }
    }
    excLds = std::max(excLds, tileExcCount[t]);
  }
  if (words > 0xFFFFFFFFull || excEntries > 0xFFFFFFFFull) {
    sb_free(lanes), sb_free(dExc), sb_free(dClassKeys);
    sb_free(m->classDict), m->classDict = nullptr;
    return giveUp();
  }
  uint32_t* dOff   = (uint32_t*)upload(chunkOff.data(), chunkOff.size() * sizeof(uint32_t));
  uint32_t* dFlags = (uint32_t*)upload(chunkFlags.data(), chunkFlags.size() * sizeof(uint32_t));
  const size_t streamBytes = (size_t)words * sizeof(uint32_t) + 1024;          // slack: clamped reads
  const size_t excBytes    = ((size_t)excEntries + 520) * sizeof(PatEntry);     // slack: 2 x 256 unconditional reads
  HIP_CHECK(hipMalloc(&m->jcodes, streamBytes));
  HIP_CHECK(hipMalloc(&m->excRows, excBytes));
  HIP_CHECK(hipMemsetAsync(m->jcodes, 0, streamBytes, g.stream));
  HIP_CHECK(hipMemsetAsync(m->excRows, 0, excBytes, g.stream));
  hipLaunchKernelGGL(pat_compact_k, dim3(nTiles), dim3(256), 0, g.stream, m->pmeta, lanes, m->nChunks, dOff,
      dFlags, dExc, m->rowBase, m->tileClass, m->classDict, m->jcodes, m->excRows);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  sb_free(lanes), sb_free(dExc), sb_free(dOff), sb_free(dFlags), sb_free(dClassKeys);
  if (rowPats.empty()) rowPats.push_back(PatEntry{ 0.0, 0u, 0u });
  m->rowPats     = (PatEntry*)upload(rowPats.data(), rowPats.size() * sizeof(PatEntry));
  m->nRowPats    = (uint32_t)patIndex.size();
  m->nPatClasses = (uint32_t)classes.size();
  m->patDict     = anyL ? 256u : 0u;
  m->patExcLds   = excLds;
  // one header per tile: class, chunk positions / widths / row patterns, the first segments
  std::vector<uint32_t> segPtr(nTiles + 1);
  sb_d2h(segPtr.data(), m->tileSegPtr, segPtr.size() * sizeof(uint32_t));
  const size_t nSegs = segPtr[nTiles];
  std::vector<TileSeg> segs(std::max<size_t>(nSegs, 1));
  if (nSegs) sb_d2h(segs.data(), m->tileSegs, nSegs * sizeof(TileSeg));
  std::vector<TileHdr> hdrs(nTiles);
  for (uint32_t t = 0; t < nTiles; t++) {
    TileHdr& h = hdrs[t];
    memset(&h, 0, sizeof h);
    h.tile = t;
    h.cls = tileClass[t], h.nseg = segPtr[t + 1] - segPtr[t], h.segPtr = segPtr[t], h.win = 1;
    h.excStart = tileExcStart[t], h.excCount = tileExcCount[t];
    for (uint32_t w = 0; w < 4; w++) {
      const uint32_t c = t * 4 + w;
      if (c >= m->nChunks) continue;
      h.off[w] = chunkOff[c], h.len[w] = chunkFlags[c], h.rowPat[w] = chunkPat[c];
      if (chunkFlags[c] & PAT_UNIFORM) h.exc[w][0] = exc[2 * (size_t)c], h.exc[w][1] = exc[2 * (size_t)c + 1];
    }
    for (uint32_t s = 0; s < PAT_INLINE_SEGS; s++) h.seg[s][1] = 0xFFFFFFFFu;
    h.winInline = 1;
    // simple window: <= 6 segments which, longest first, are 3 x <= 768 and 3 x <= 256 entries
    std::vector<TileSeg> ts(segs.begin() + segPtr[t], segs.begin() + segPtr[t] + h.nseg);
    std::stable_sort(ts.begin(), ts.end(), [](const TileSeg& a, const TileSeg& b) { return a.len > b.len; });
    bool simple = h.nseg <= PAT_INLINE_SEGS;
    for (uint32_t s = 0; s < h.nseg && simple; s++) simple = ts[s].len <= (s < 3 ? 768u : 256u);
    for (uint32_t s = 0; s < h.nseg; s++) {
      const TileSeg& sg = simple ? ts[s] : segs[segPtr[t] + s]; // slot order unless simple
      if (s < PAT_INLINE_SEGS) h.seg[s][0] = sg.col, h.seg[s][1] = sg.lds, h.seg[s][2] = sg.len;
    }
    for (uint32_t s = 0; s < h.nseg; s++) {
      const TileSeg& sg = segs[segPtr[t] + s];
      if (s < PAT_INLINE_SEGS) h.winInline = sg.lds + sg.len;
      h.win = sg.lds + sg.len;
    }
    h.flags = simple ? PAT_SIMPLE_WINDOW : 0u;
  }
  // tiles whose window holds a halo column (>= nr) go last: the interior part of the product
  // does not have to wait for the halo exchange (loop_body)
  m->patInterior = nTiles;
  if (m->nc > m->nr) {
    auto touches_halo = [&](const TileHdr& h) {
      for (uint32_t s2 = 0; s2 < h.nseg; s2++) {
        const TileSeg& sg = segs[h.segPtr + s2];
        if (sg.col + sg.len > m->nr) return true;
      }
      return false;
    };
    auto mid = std::stable_partition(hdrs.begin(), hdrs.end(), [&](const TileHdr& h) { return !touches_halo(h); });
    m->patInterior = (uint32_t)(mid - hdrs.begin());
  }
  m->tileHdrs = (TileHdr*)upload(hdrs.data(), hdrs.size() * sizeof(TileHdr));
  if (getenv("SB_PACK_REPORT")) {
    size_t nSimple = 0;
    for (const TileHdr& h : hdrs) nSimple += h.flags & PAT_SIMPLE_WINDOW;
    fprintf(stderr, "sbhip pack: %u tiles (%u interior, %zu simple windows, max %u entries), %u classes, %u/%u U chunks, "
                    "%zu row patterns (%zu entries), %llu exception entries (max %u per tile), %llu code words\n",
        nTiles, m->patInterior, nSimple, m->ldsWindow, m->nPatClasses, m->nUniformChunks, m->nChunks, patIndex.size(),
        rowPats.size(), (unsigned long long)excEntries, excLds, (unsigned long long)words);
  }
  m->patBytes = (double)words * 4.0 + 16.0 * (double)excEntries + 2.0 * 64.0 * m->nChunks + 16.0 * nSegs +
                (double)sizeof(TileHdr) * nTiles + (anyL ? 4096.0 * classes.size() : 0.0) + 16.0 * rowPats.size();
  // Default kernel: the pattern kernel once the matrix is more than one round of resident
  // workgroups (8 per CU); below that everything is one dependent-latency chain and the
  // level-3 kernel's is shorter (64^3: 46.7k vs 43.2k CG it/s; 96^3: 22.8k vs 26.3k).
  // sb_matrix_use_packed(m, 3) selects it regardless.
  m->usePacked = nTiles > (uint32_t)g.prop.multiProcessorCount * 8u ? 3 : 2;
}

sb_matrix* sb_scs_upload(uint32_t nr, uint32_t nc, uint32_t C, uint32_t sigma, uint32_t nChunks,
    uint32_t nElems, const uint32_t* chunkPtr, const uint32_t* chunkLens, const uint32_t* colInd,
    const double* val, const uint32_t* oldToNewPerm, const uint32_t* newToOldPerm)
{
  need_init();
  if (C == 0) SB_FATAL("SCS chunk height C must be >= 1");
  if ((uint64_t)nChunks * C < nr) SB_FATAL("SCS nChunks*C < nr");
  if (chunkPtr[nChunks] != nElems) SB_FATAL("SCS chunkPtr[nChunks] != nElems");
  for (uint32_t c = 0; c < nChunks; c++)
    if (chunkPtr[c + 1] - chunkPtr[c] != chunkLens[c] * C)
      SB_FATAL("SCS chunk %u: chunkPtr/chunkLens inconsistent", c);
  for (uint32_t k = 0; k < nElems; k++)
    if (colInd[k] >= nc) SB_FATAL("SCS colInd[%u]=%u out of range (nc=%u)", k, colInd[k], nc);
  sb_matrix* m = new sb_matrix();
  m->fmt = 1, m->nr = nr, m->nc = nc, m->C = C, m->sigma = sigma, m->nChunks = nChunks;
  m->nElems = nElems, m->nrPadded = nChunks * C, m->nnz = nElems;
  int permuted = 0;
  if (oldToNewPerm)
    for (uint32_t i = 0; i < nr; i++) {
      if (oldToNewPerm[i] >= nr) SB_FATAL("SCS oldToNewPerm[%u]=%u out of range", i, oldToNewPerm[i]);
      if (oldToNewPerm[i] != i) permuted = 1;
    }
  if (permuted && !newToOldPerm) SB_FATAL("SCS permuted matrix needs newToOldPerm");
  m->permuted  = permuted;
  m->chunkPtr  = (uint32_t*)upload(chunkPtr, ((size_t)nChunks + 1) * sizeof(uint32_t));
  m->chunkLens = (uint32_t*)upload(chunkLens, (size_t)nChunks * sizeof(uint32_t));
  // SCS_SLACK zeroed elements behind the data: the pipelined kernel prefetches up to
  // U-1 columns past a chunk's end
  HIP_CHECK(hipMalloc(&m->colInd, ((size_t)nElems + SCS_SLACK) * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&m->val, ((size_t)nElems + SCS_SLACK) * sizeof(double)));
  HIP_CHECK(hipMemset(m->colInd + nElems, 0, SCS_SLACK * sizeof(uint32_t)));
  HIP_CHECK(hipMemset(m->val + nElems, 0, SCS_SLACK * sizeof(double)));
  if (nElems) {
    HIP_CHECK(hipMemcpy(m->colInd, colInd, (size_t)nElems * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->val, val, (size_t)nElems * sizeof(double), hipMemcpyHostToDevice));
  }
  if (permuted) {
    m->oldToNew = (uint32_t*)upload(oldToNewPerm, (size_t)nr * sizeof(uint32_t));
    m->newToOld = (uint32_t*)upload(newToOldPerm, (size_t)nr * sizeof(uint32_t));
    hipLaunchKernelGGL(remap_cols_k, dim3(stream_grid(nElems, 256)), dim3(256), 0, g.stream, nElems,
        nr, m->oldToNew, m->colInd);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(g.stream));
  }
  build_packed(m, val, oldToNewPerm);
  build_lds_windows(m, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
  build_patterns(m);
  return m;
}

// CRS: a device-private Sell-64-1 mirror that exists only for its pattern levels (pack.hip.h).
// Row sums are taken left to right exactly as src/matrix-CRS.c:46-65 does; the SKIPPAD kernel
// does not add the mirror's padding, so the result is the CRS loop's bit for bit.  Kept only
// when the pattern levels could be built; the native CRS kernel stays selectable
// (sb_matrix_use_packed(m, 0)) and is the default for small matrices.
static void build_crs_mirror(sb_matrix* m, const uint32_t* rowPtr, const uint32_t* colInd, const double* val)
{
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 5) < 4 || m->nr == 0 || m->nnz == 0) return;
  const uint32_t nr = m->nr, nChunks = (nr + 63) / 64;
  std::vector<uint32_t> chunkLens(nChunks, 0), chunkPtr(nChunks + 1, 0);
  for (uint32_t i = 0; i < nr; i++) chunkLens[i / 64] = std::max(chunkLens[i / 64], rowPtr[i + 1] - rowPtr[i]);
  uint64_t total = 0;
  for (uint32_t c = 0; c < nChunks; c++) {
    chunkPtr[c] = (uint32_t)total;
    total += (uint64_t)chunkLens[c] * 64;
  }
  if (total > 0xFFFFFFFFull) return;
  chunkPtr[nChunks] = (uint32_t)total;
  std::vector<uint32_t> scol(total, 0u);
  std::vector<double> sval(total, 0.0);
  for (uint32_t i = 0; i < nr; i++) {
    const size_t at = (size_t)chunkPtr[i / 64] + (i % 64);
    for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; j++) {
      unsigned long long bits;
      memcpy(&bits, val + j, 8);
      // padding is (column 0, +0.0): a stored +0.0 at column 0 would be indistinguishable from it
      if (colInd[j] == 0 && bits == 0) return;
      scol[at + (size_t)(j - rowPtr[i]) * 64] = colInd[j];
      sval[at + (size_t)(j - rowPtr[i]) * 64] = val[j];
    }
  }
  sb_matrix* mm = sb_scs_upload(nr, m->nc, 64, 1, nChunks, (uint32_t)total, chunkPtr.data(), chunkLens.data(),
      scol.data(), sval.data(), nullptr, nullptr);
  if (mm->nPatClasses == 0) { // no repeating patterns: the native kernel it is
    sb_matrix_free(mm);
    return;
  }
  // only the pattern levels are used (the other SCS kernels add the padding)
  sb_free(mm->val), sb_free(mm->colInd), sb_free(mm->pidx), sb_free(mm->pcodes), sb_free(mm->pslots);
  mm->val = nullptr, mm->colInd = nullptr, mm->pidx = nullptr, mm->pcodes = nullptr, mm->pslots = nullptr;
  m->mirror    = mm;
  m->usePacked = mm->usePacked == 3 ? 3 : 0; // the same size rule as for SCS matrices
}

void sb_matrix_free(sb_matrix* m)
{
  if (!m) return;
  sb_free(m->rowPtr), sb_free(m->rowBlocks), sb_free(m->chunkPtr), sb_free(m->chunkLens);
  sb_free(m->oldToNew), sb_free(m->newToOld), sb_free(m->colInd), sb_free(m->val);
  sb_free(m->pmeta), sb_free(m->pidx), sb_free(m->pcodes), sb_free(m->pdict);
  sb_free(m->tileSegPtr), sb_free(m->tileSegs), sb_free(m->pslots);
  sb_free(m->rowBase), sb_free(m->tileClass), sb_free(m->jcodes), sb_free(m->classDict), sb_free(m->tileHdrs), sb_free(m->rowPats), sb_free(m->excRows);
  if (m->mirror) sb_matrix_free(m->mirror);
  delete m;
}

int sb_matrix_pack_level(const sb_matrix* m) { return m->packLevel; }
void sb_matrix_use_packed(sb_matrix* m, int mode)
{ // 0 reference-layout stream, 1 packed stream + gathers through the cache, 2 packed + LDS window,
  // 3 pattern codes + LDS window; a mode the matrix does not have falls to the next lower one
  if (m->fmt == 0) m->usePacked = mode >= 3 && m->mirror ? 3 : 0;
  else if (mode >= 3 && m->nPatClasses) m->usePacked = 3;
  else if (mode >= 2 && m->ldsWindow) m->usePacked = 2;
  else if (mode >= 1 && m->packLevel) m->usePacked = 1;
  else m->usePacked = 0;
}
int sb_matrix_packed_mode(const sb_matrix* m) { return m->usePacked; }
// the matrix whose pattern levels serve m: m itself (SCS) or its private mirror (CRS)
static const sb_matrix* pat_of(const sb_matrix* m) { return m->fmt == 0 && m->mirror ? m->mirror : m; }
uint32_t sb_matrix_lds_window(const sb_matrix* m) { return pat_of(m)->ldsWindow; }
uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }
uint32_t sb_matrix_row_patterns(const sb_matrix* m, uint32_t* uniformChunks)
{
  if (uniformChunks) *uniformChunks = pat_of(m)->nUniformChunks;
  return pat_of(m)->nRowPats;
}
double sb_matrix_stream_bytes(const sb_matrix* m)
{ // bytes the SELECTED SpMV kernel moves per launch (matrix stream + x once + y once)
  if (m->fmt == 0 && m->usePacked == 3) return m->mirror->patBytes + 8.0 * m->mirror->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked == 3) return m->patBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked == 2) return m->slotBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked) return m->packedBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  return sb_matrix_spmv_bytes(m);
}
uint32_t sb_matrix_nr(const sb_matrix* m) { return m->nr; }
uint32_t sb_matrix_nc(const sb_matrix* m) { return m->nc; }
int sb_matrix_is_permuted(const sb_matrix* m) { return m->permuted; }
double sb_matrix_spmv_bytes(const sb_matrix* m)
{
  if (m->fmt == 0)
    return 12.0 * m->nnz + 4.0 * ((double)m->nr + 1) + 8.0 * m->nr + 8.0 * m->nc;
  return 12.0 * m->nElems + 8.0 * m->nChunks + 8.0 * m->nrPadded + 8.0 * m->nc;
}

// ===========================================================================
// kernels
// ===========================================================================
static int g_scs_unroll = -1;
static int g_scs_nt     = -1;
static int g_scs_xcd    = 1;

// dotPartials != NULL: fuse the level-0 partials of p.Ap into the SpMV (SCS C=64 only)
// part: 0 the whole product; 1 / 2 its interior / halo-touching tiles (spmv_can_split only)
static bool spmv_uses_patterns(const sb_matrix* m)
{
  return m->usePacked == 3 && (m->fmt == 0 ? m->mirror != nullptr : m->C == 64);
}
static bool spmv_can_split(const sb_matrix* m)
{
  const sb_matrix* pm = pat_of(m);
  return spmv_uses_patterns(m) && pm->patInterior > 0 && pm->patInterior < (pm->nChunks + 3) / 4;
}
static void launch_pat(const sb_matrix* pm, bool skipPad, const double* x, double* y, double* dotPartials,
    const int* stop, int part);

static void launch_spmv(const sb_matrix* m, const double* x, double* y, double* dotPartials,
    const int* stop, int part = 0)
{
  const bool dot = dotPartials != nullptr;
  if (m->nr == 0) return;
  if (part != 0 && !spmv_can_split(m)) SB_FATAL("this SpMV kernel cannot be launched in parts");
  if (m->fmt == 0 && spmv_uses_patterns(m)) {
    launch_pat(m->mirror, true, x, y, dotPartials, stop, part);
  } else if (m->fmt == 0) {
    if (dot) SB_FATAL("fused dot needs the pattern kernel (SCS C=64, or CRS through its mirror)");
    const uint32_t per = (m->nRowBlocks + 7) / 8;
    hipLaunchKernelGGL(spmv_crs_stream, dim3(per * 8), dim3(CRS_THREADS), 0, g.stream, m->rowBlocks,
        m->rowPtr, m->colInd, m->val, x, y, m->nRowBlocks, per, stop);
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
    if (m->usePacked == 3) {
      launch_pat(m, false, x, y, dotPartials, stop, part);
    } else if (m->usePacked == 2) {
      const size_t shmem = (256 + (size_t)m->ldsWindow) * sizeof(double);
#define LDS_LAUNCH(DI, DO)                                                                                   \
  hipLaunchKernelGGL((spmv_scs64_lds<DI, DO>), grid, block, shmem, g.stream, m->pmeta, m->pslots, m->pcodes, \
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
  hipLaunchKernelGGL((spmv_scs64_packed<DI, DO>), grid, block, 0, g.stream, m->pmeta, m->pidx, m->pcodes, \
      m->pdict, m->chunkPtr, m->val, x, y, m->nr, m->nChunks, per, m->padCol, dotPartials, stop)
      if (m->nDict > 0) {
        if (dot) PK_LAUNCH(true, true);
        else PK_LAUNCH(true, false);
      } else {
        if (dot) PK_LAUNCH(false, true);
        else PK_LAUNCH(false, false);
      }
#undef PK_LAUNCH
    } else {
#define SCS_LAUNCH(U, D, N)                                                                      \
  hipLaunchKernelGGL((spmv_scs64<U, D, N>), grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, \
      m->colInd, m->val, x, y, m->nr, m->nChunks, per, dotPartials, stop)
#define SCS_PICK(U)                                                                           \
  do {                                                                                        \
    if (dot) { if (g_scs_nt) SCS_LAUNCH(U, true, true); else SCS_LAUNCH(U, true, false); }    \
    else { if (g_scs_nt) SCS_LAUNCH(U, false, true); else SCS_LAUNCH(U, false, false); }      \
  } while (0)
      switch (g_scs_unroll) {
      case 1: SCS_PICK(1); break;
      case 2: SCS_PICK(2); break;
      case 8: SCS_PICK(8); break;
      case 9: SCS_PICK(9); break;
      default: SCS_PICK(4); break;
      }
#undef SCS_PICK
#undef SCS_LAUNCH
    }
  } else {
    if (dot) SB_FATAL("fused dot is an SCS C=64 feature");
    hipLaunchKernelGGL(spmv_scs_generic, dim3((m->nrPadded + 255) / 256), dim3(256), 0, g.stream,
        m->chunkPtr, m->chunkLens, m->colInd, m->val, x, y, m->nr, m->nrPadded, m->C, stop);
  }
  HIP_CHECK(hipGetLastError());
}

static void launch_pat(const sb_matrix* pm, bool skipPad, const double* x, double* y, double* dotPartials,
    const int* stop, int part)
{
  const bool dot         = dotPartials != nullptr;
  const uint32_t nBlocks = (pm->nChunks + 3) / 4;
  const size_t shmem = ((size_t)pm->patDict + pm->patExcLds + 8) * sizeof(PatEntry) + (size_t)pm->ldsWindow * sizeof(double);
  if (!stop) stop = zero_flag();
  const uint32_t first = part == 2 ? pm->patInterior : 0u;
  const uint32_t count = part == 1 ? pm->patInterior : part == 2 ? nBlocks - pm->patInterior : nBlocks;
  const uint32_t pper  = g_scs_xcd ? (count + 7) / 8 : 0;
  const dim3 pgrid(g_scs_xcd ? pper * 8 : count), block(256);
#define PAT_LAUNCH(DO, SK)                                                                                       \
  hipLaunchKernelGGL((spmv_scs64_pat<DO, SK>), pgrid, block, shmem, g.stream, pm->tileHdrs, pm->jcodes, pm->rowBase, \
      pm->classDict, pm->rowPats, pm->excRows, pm->tileSegs, x, y, pm->nr, pm->nChunks, first, count, pper,      \
      pm->padCol, pm->patDict, pm->patExcLds, dotPartials, stop)
  if (skipPad) {
    if (dot) PAT_LAUNCH(true, true);
    else PAT_LAUNCH(false, true);
  } else {
    if (dot) PAT_LAUNCH(true, false);
    else PAT_LAUNCH(false, false);
  }
#undef PAT_LAUNCH
  HIP_CHECK(hipGetLastError());
}

void sb_spmv_native(const sb_matrix* m, const double* x, double* y)
{
  need_init();
  launch_spmv(m, x, y, nullptr, nullptr);
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
  else hipLaunchKernelGGL((dot_spans_k<3>), grid, block, 0, g.stream, n, a, b, x, r, S, partials, stop);
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
// communicator + halo
// ===========================================================================
void sb_comm_unique_id(void* id_out)
{
  rccl_open();
  ncclUniqueId id;
  RCCL_CHECK(rccl.GetUniqueId(&id));
  memcpy(id_out, &id, SB_UNIQUE_ID_BYTES);
}

void sb_comm_init(int rank, int size, const void* idbytes)
{
  need_init();
  if (g.comm) SB_FATAL("communicator already initialised");
  if (size < 1 || rank < 0 || rank >= size) SB_FATAL("bad rank %d / size %d", rank, size);
  g.rank = rank, g.size = size;
  // serial: every comm call degrades to a no-op (src/comm.c:404-411).  SB_FORCE_RCCL=1
  // builds a 1-rank RCCL communicator anyway, so the multi-rank kernel sequence and the
  // RCCL bindings can be exercised on a single GPU.
  if (size == 1 && !getenv("SB_FORCE_RCCL")) return;
  rccl_open();
  ncclUniqueId id;
  memcpy(&id, idbytes, SB_UNIQUE_ID_BYTES);
  RCCL_CHECK(rccl.CommInitRank(&g.comm, size, id, rank));
  if (size > 1) { // peer-mapped buffers for the in-kernel all-reduce; the handles travel over RCCL
    unsigned char mine[SB_P2P_HANDLE_BYTES], all[SB_P2P_HANDLE_BYTES * P2P_MAX];
    const int have = size <= P2P_MAX && sb_comm_p2p_handle(mine);
    if (!have) memset(mine, 0, sizeof mine);
    if (size <= P2P_MAX) {
      sb_comm_allgather_bytes(mine, SB_P2P_HANDLE_BYTES, all);
      sb_comm_p2p_open(have ? all : nullptr);
    }
  }
}

void sb_comm_init_transport(int rank, int size, const sb_transport* t)
{
  need_init();
  if (g.comm || g.hasXport) SB_FATAL("communicator already initialised");
  if (size < 1 || rank < 0 || rank >= size || !t || !t->allreduce || !t->neighbour_exchange)
    SB_FATAL("bad transport / rank %d / size %d", rank, size);
  g.rank = rank, g.size = size, g.xport = *t, g.hasXport = true;
}

// ---- in-kernel all-reduce over peer-mapped memory: set-up ---------------------------------
static void p2p_release()
{
  for (int r = 0; r < P2P_MAX; r++) {
    if (g.p2pPeer[r] && g.p2pPeer[r] != (void*)g.p2pBuf) (void)hipIpcCloseMemHandle(g.p2pPeer[r]);
    g.p2pPeer[r] = nullptr;
  }
  if (g.p2pBuf) (void)hipFree(g.p2pBuf);
  if (g.p2pView) (void)hipFree(g.p2pView);
  g.p2pBuf = nullptr, g.p2pView = nullptr, g.p2pOn = false, g.p2pSeq = 0;
}

int sb_comm_p2p_handle(unsigned char* handle_out)
{
  need_init();
  const char* env = getenv("SB_P2P");
  if (env && atoi(env) == 0) return 0;
  static_assert(sizeof(hipIpcMemHandle_t) <= SB_P2P_HANDLE_BYTES, "IPC handle size");
  if (!g.p2pBuf) {
    void* buf = nullptr; // fine-grained: coherent between GPUs while kernels are running
    if (hipExtMallocWithFlags(&buf, 2 * P2P_MAX * sizeof(P2PSlot), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    g.p2pBuf = (P2PSlot*)buf;
    HIP_CHECK(hipMemset(g.p2pBuf, 0, 2 * P2P_MAX * sizeof(P2PSlot)));
    HIP_CHECK(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, g.p2pBuf) != hipSuccess) {
    (void)hipGetLastError();
    p2p_release();
    return 0;
  }
  memset(handle_out, 0, SB_P2P_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof h);
  return 1;
}

int sb_comm_p2p_open(const unsigned char* all_handles)
{
  need_init();
  if (!multi_rank()) return 0;
  int ok = all_handles != nullptr && g.p2pBuf != nullptr && g.size <= P2P_MAX;
  unsigned char zero[SB_P2P_HANDLE_BYTES] = { 0 };
  for (int r = 0; ok && r < g.size; r++) {
    const unsigned char* hb = all_handles + (size_t)r * SB_P2P_HANDLE_BYTES;
    if (memcmp(hb, zero, SB_P2P_HANDLE_BYTES) == 0) ok = 0; // that rank has none
    else if (r == g.rank) g.p2pPeer[r] = g.p2pBuf;
    else {
      hipIpcMemHandle_t h;
      memcpy(&h, hb, sizeof h);
      if (hipIpcOpenMemHandle(&g.p2pPeer[r], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        g.p2pPeer[r] = nullptr;
        ok = 0;
      }
    }
  }
  // every rank must come to the same decision.  Round 1 (the established transport): did everybody
  // map everybody?  Round 2: one in-kernel exchange, checked, and agreed on over the transport again.
  double* d = (double*)sb_malloc(4 * sizeof(double));
  auto agree = [&](int mine) {
    const double v = mine ? 1.0 : 0.0;
    sb_h2d(d, &v, sizeof v);
    sb_comm_reduction(d, 1);
    double sum = 0.0;
    sb_d2h(&sum, d, sizeof sum);
    return sum == (double)g.size;
  };
  bool on = agree(ok);
  if (on) {
    P2PView view;
    memset(&view, 0, sizeof view);
    view.rank = g.rank, view.size = g.size;
    for (int r = 0; r < g.size; r++) view.peer[r] = (P2PSlot*)g.p2pPeer[r];
    HIP_CHECK(hipMalloc(&g.p2pView, sizeof view));
    HIP_CHECK(hipMemcpy(g.p2pView, &view, sizeof view, hipMemcpyHostToDevice));
    int* err = (int*)(d + 2);
    HIP_CHECK(hipMemset(d, 0, 4 * sizeof(double)));
    hipLaunchKernelGGL(p2p_selftest_k, dim3(1), dim3(64), 0, g.stream, (const P2PView*)g.p2pView, ++g.p2pSeq,
        (double)(g.rank + 1), d + 1, err);
    HIP_CHECK(hipGetLastError());
    double got = 0.0;
    int e      = 0;
    sb_d2h(&got, d + 1, sizeof got);
    sb_d2h(&e, err, sizeof e);
    on = agree(!e && got == 0.5 * g.size * (g.size + 1));
  }
  sb_free(d);
  if (!on) p2p_release();
  g.p2pOn = on;
  if (getenv("SB_PACK_REPORT") || getenv("SB_P2P_REPORT"))
    fprintf(stderr, "sbhip comm: rank %d/%d in-kernel all-reduce over peer-mapped memory: %s\n", g.rank, g.size,
        on ? "on" : "off (RCCL / transport all-reduce)");
  return on ? 1 : 0;
}

int sb_comm_p2p_enabled(void) { return g.p2pOn ? 1 : 0; }

void sb_comm_finalize(void)
{
  if (g.init) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    p2p_release();
  }
  g.hasXport = false;
  if (g.comm) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    RCCL_CHECK(rccl.CommDestroy(g.comm));
    g.comm = nullptr;
  }
  g.rank = 0, g.size = 1;
}

int sb_comm_rank(void) { return g.rank; }
int sb_comm_size(void) { return g.size; }

void sb_comm_reduction(double* v_dev, int op)
{
  need_init();
  if (g.hasXport) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    g.xport.allreduce(g.xport.ctx, v_dev, op);
    return;
  }
  if (!g.comm) return;
  RCCL_CHECK(rccl.AllReduce(v_dev, v_dev, 1, ncclFloat64_, op == 0 ? ncclMax_ : ncclSum_, g.comm,
      g.stream));
}

void sb_comm_allgather_bytes(const void* mine_host, int nbytes, void* all_host)
{
  need_init();
  if (!g.comm) {
    memcpy(all_host, mine_host, (size_t)nbytes);
    return;
  }
  char *dsend = nullptr, *drecv = nullptr;
  HIP_CHECK(hipMalloc(&dsend, (size_t)nbytes + 8));
  HIP_CHECK(hipMalloc(&drecv, (size_t)nbytes * g.size + 8));
  HIP_CHECK(hipMemcpyAsync(dsend, mine_host, (size_t)nbytes, hipMemcpyHostToDevice, g.stream));
  RCCL_CHECK(rccl.AllGather(dsend, drecv, (size_t)nbytes, ncclInt8_, g.comm, g.stream));
  HIP_CHECK(hipMemcpyAsync(all_host, drecv, (size_t)nbytes * g.size, hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  HIP_CHECK(hipFree(dsend));
  HIP_CHECK(hipFree(drecv));
}

void sb_comm_alltoallv_ints(const int* sendbuf, const int* sendcounts, const int* sdispls, int* recvbuf,
    const int* recvcounts, const int* rdispls)
{
  need_init();
  const int me = g.rank;
  if (!g.comm) {
    memcpy(recvbuf + rdispls[0], sendbuf + sdispls[0], (size_t)sendcounts[0] * sizeof(int));
    return;
  }
  size_t ns = 0, nr = 0;
  for (int r = 0; r < g.size; r++) {
    if ((size_t)(sdispls[r] + sendcounts[r]) > ns) ns = (size_t)(sdispls[r] + sendcounts[r]);
    if ((size_t)(rdispls[r] + recvcounts[r]) > nr) nr = (size_t)(rdispls[r] + recvcounts[r]);
  }
  int *dsend = nullptr, *drecv = nullptr;
  HIP_CHECK(hipMalloc(&dsend, (ns + 2) * sizeof(int)));
  HIP_CHECK(hipMalloc(&drecv, (nr + 2) * sizeof(int)));
  HIP_CHECK(hipMemcpyAsync(dsend, sendbuf, ns * sizeof(int), hipMemcpyHostToDevice, g.stream));
  RCCL_CHECK(rccl.GroupStart());
  for (int r = 0; r < g.size; r++) {
    if (r == me) continue;
    if (sendcounts[r])
      RCCL_CHECK(rccl.Send(dsend + sdispls[r], (size_t)sendcounts[r], ncclInt32_, r, g.comm, g.stream));
    if (recvcounts[r])
      RCCL_CHECK(rccl.Recv(drecv + rdispls[r], (size_t)recvcounts[r], ncclInt32_, r, g.comm, g.stream));
  }
  RCCL_CHECK(rccl.GroupEnd());
  if (sendcounts[me])
    HIP_CHECK(hipMemcpyAsync(drecv + rdispls[me], dsend + sdispls[me], (size_t)sendcounts[me] * sizeof(int),
        hipMemcpyDeviceToDevice, g.stream));
  HIP_CHECK(hipMemcpyAsync(recvbuf, drecv, nr * sizeof(int), hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  HIP_CHECK(hipFree(dsend));
  HIP_CHECK(hipFree(drecv));
}

void sb_comm_barrier(void)
{
  need_init();
  if (g.comm) {
    HIP_CHECK(hipMemsetAsync(g.scalar, 0, sizeof(double), g.stream));
    sb_comm_reduction(g.scalar, 1);
  }
  HIP_CHECK(hipStreamSynchronize(g.stream));
}

sb_halo* sb_halo_create(uint32_t nr, int outdegree, const int* destinations, const int* sendCounts,
    const int* sdispls, int indegree, const int* sources, const int* recvCounts, const int* rdispls,
    const int* elementsToSend, int totalSendCount, int externalCount, const uint32_t* oldToNewPerm)
{
  need_init();
  sb_halo* h        = new sb_halo();
  h->nr             = nr;
  h->outdegree      = outdegree;
  h->indegree       = indegree;
  h->totalSend      = totalSendCount;
  h->externalCount  = externalCount;
  h->destinations.assign(destinations, destinations + outdegree);
  h->sendCounts.assign(sendCounts, sendCounts + outdegree);
  h->sdispls.assign(sdispls, sdispls + outdegree);
  h->sources.assign(sources, sources + indegree);
  h->recvCounts.assign(recvCounts, recvCounts + indegree);
  h->rdispls.assign(rdispls, rdispls + indegree);
  int sum = 0;
  for (int i = 0; i < outdegree; i++) {
    if (sdispls[i] != sum) SB_FATAL("halo: sdispls must be the prefix sums of sendCounts");
    sum += sendCounts[i];
  }
  if (sum != totalSendCount) SB_FATAL("halo: totalSendCount mismatch");
  sum = 0;
  for (int i = 0; i < indegree; i++) {
    if (rdispls[i] != sum) SB_FATAL("halo: rdispls must be the prefix sums of recvCounts");
    sum += recvCounts[i];
  }
  if (sum != externalCount) SB_FATAL("halo: externalCount mismatch");
  std::vector<uint32_t> idx((size_t)totalSendCount);
  for (int i = 0; i < totalSendCount; i++) {
    if (elementsToSend[i] < 0 || (uint32_t)elementsToSend[i] >= nr)
      SB_FATAL("halo: elementsToSend[%d]=%d out of range", i, elementsToSend[i]);
    idx[i] = oldToNewPerm ? oldToNewPerm[elementsToSend[i]] : (uint32_t)elementsToSend[i];
  }
  h->packIdx = (uint32_t*)upload(idx.data(), idx.size() * sizeof(uint32_t));
  HIP_CHECK(hipMalloc(&h->sendBuf, ((size_t)totalSendCount + 1) * sizeof(double)));
  return h;
}

void sb_halo_free(sb_halo* h)
{
  if (!h) return;
  sb_free(h->packIdx), sb_free(h->sendBuf);
  delete h;
}

static void halo_exchange(sb_halo* h, double* x, const int* stop, hipStream_t stream = nullptr)
{
  if (!h || g.size == 1) return;
  if (!stream) stream = g.stream;
  if (h->totalSend) {
    hipLaunchKernelGGL(gather_k, dim3(stream_grid(h->totalSend, 256)), dim3(256), 0, stream,
        (uint32_t)h->totalSend, h->packIdx, x, h->sendBuf, stop);
    HIP_CHECK(hipGetLastError());
  }
  // neighbour all-to-all (MPI_Neighbor_alltoallv, src/comm.c:640-648) as one
  // RCCL group of point-to-point transfers over xGMI, received straight into the
  // tail of x (no unpack), stream-ordered.
  if (g.hasXport) {
    HIP_CHECK(hipStreamSynchronize(stream));
    g.xport.neighbour_exchange(g.xport.ctx, h->sendBuf, h->outdegree, h->destinations.data(),
        h->sendCounts.data(), h->sdispls.data(), x + h->nr, h->indegree, h->sources.data(),
        h->recvCounts.data(), h->rdispls.data());
    return;
  }
  RCCL_CHECK(rccl.GroupStart());
  for (int i = 0; i < h->outdegree; i++)
    RCCL_CHECK(rccl.Send(h->sendBuf + h->sdispls[i], (size_t)h->sendCounts[i], ncclFloat64_,
        h->destinations[i], g.comm, stream));
  for (int i = 0; i < h->indegree; i++)
    RCCL_CHECK(rccl.Recv(x + h->nr + h->rdispls[i], (size_t)h->recvCounts[i], ncclFloat64_,
        h->sources[i], g.comm, stream));
  RCCL_CHECK(rccl.GroupEnd());
}

void sb_halo_exchange(sb_halo* h, double* x)
{
  need_init();
  halo_exchange(h, x, nullptr);
}

// ===========================================================================
// CG
// ===========================================================================
enum { R_WAXPBY = 0, R_SPMVM = 1, R_DDOT = 2, R_COMM = 3 };

static void mark(sb_cg* s, int region)
{ // region = the region that ENDS here (-1: start marker)
  if (!s->timing) return;
  if (s->evUsed == s->ev.size()) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    s->ev.push_back(e);
    s->evRegion.push_back(-1);
  }
  s->evRegion[s->evUsed] = region;
  HIP_CHECK(hipEventRecord(s->ev[s->evUsed], g.stream));
  s->evUsed++;
}

sb_cg* sb_cg_create(const sb_matrix* m, sb_halo* halo, const double* b_host, const double* xexact_host)
{
  need_init();
  sb_cg* s = new sb_cg();
  s->A = m, s->halo = halo, s->nr = m->nr, s->nc = m->nc;
  if (halo && halo->nr != m->nr) SB_FATAL("halo plan and matrix disagree on nr");
  if (halo && m->nr + (uint32_t)halo->externalCount != m->nc) SB_FATAL("halo externalCount != nc-nr");
  const size_t nb = (size_t)m->nr * sizeof(double);
  s->r  = (double*)sb_malloc(nb);
  s->Ap = (double*)sb_malloc(nb);
  s->x  = (double*)sb_malloc(nb);
  s->b  = (double*)sb_malloc(nb);
  s->p  = (double*)sb_malloc((size_t)m->nc * sizeof(double)); // nc = nr + externals (src/CGSolver.c:70)
  s->xexact = xexact_host ? (double*)sb_malloc(nb) : nullptr;
  double* tmp = scratch_ws(0, m->nr);
  sb_h2d(tmp, b_host, nb);
  sb_permute(m, tmp, s->b);
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (xexact_host) {
    sb_h2d(tmp, xexact_host, nb);
    sb_permute(m, tmp, s->xexact);
    HIP_CHECK(hipStreamSynchronize(g.stream));
  }
  s->S         = (CgScalars*)sb_malloc(sizeof(CgScalars));
  s->nPartials = (m->nr + 255) / 256;
  // level-0 partials: 4 per 256 rows; the tail beyond the last chunk stays +0.0
  s->partials = (double*)sb_malloc((4 * (size_t)s->nPartials + 4) * sizeof(double));
  HIP_CHECK(hipMemsetAsync(s->partials, 0, (4 * (size_t)s->nPartials + 4) * sizeof(double), g.stream));
  s->hist_cap  = 0;
  s->rr_hist = s->pAp_hist = nullptr;
  s->fused      = 1;
  s->use_graph  = 0;
  s->graphReady = false;
  s->iterGraph  = nullptr;
  s->timing     = false;
  s->evUsed     = 0;
  s->loop_ms    = 0.f;
  s->spmvTiming = false;
  s->spmvEvUsed = 0;
  s->k_next     = 1;
  s->started    = false;
  HIP_CHECK(hipEventCreate(&s->evLoop0));
  HIP_CHECK(hipEventCreate(&s->evLoop1));
  for (double& v : s->region_ms) v = 0.0;
  return s;
}

void sb_cg_free(sb_cg* s)
{
  if (!s) return;
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (s->iterGraph) HIP_CHECK(hipGraphExecDestroy(s->iterGraph));
  for (hipEvent_t e : s->ev) HIP_CHECK(hipEventDestroy(e));
  for (hipEvent_t e : s->spmvEv) HIP_CHECK(hipEventDestroy(e));
  HIP_CHECK(hipEventDestroy(s->evLoop0));
  HIP_CHECK(hipEventDestroy(s->evLoop1));
  sb_free(s->r), sb_free(s->Ap), sb_free(s->x), sb_free(s->b), sb_free(s->p), sb_free(s->xexact);
  sb_free(s->S), sb_free(s->partials), sb_free(s->rr_hist), sb_free(s->pAp_hist);
  delete s;
}

static void drop_graph(sb_cg* s)
{
  if (s->iterGraph) HIP_CHECK(hipGraphExecDestroy(s->iterGraph));
  s->iterGraph = nullptr, s->graphReady = false;
}

void sb_cg_set_fused(sb_cg* s, int fused)
{
  if (s->fused != fused) drop_graph(s);
  s->fused = fused;
}
void sb_cg_set_graph(sb_cg* s, int use_graph) { s->use_graph = use_graph; }

void sb_cg_spmv_timing(sb_cg* s, int on)
{
  s->spmvTiming = on != 0;
  s->spmvEvUsed = 0;
}

double sb_cg_spmv_ms(sb_cg* s, int* launches)
{ // sum of the event-bracketed SpMV launches since sb_cg_spmv_timing(s, 1)
  HIP_CHECK(hipStreamSynchronize(g.stream));
  double total = 0.0;
  int n        = 0;
  for (size_t i = 0; i + 1 < s->spmvEvUsed; i += 2) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, s->spmvEv[i], s->spmvEv[i + 1]));
    total += ms;
    n++;
  }
  if (launches) *launches = n;
  return total;
}

void sb_cg_counters(const sb_cg* s, int out[5])
{ // stop, stop_next, iters, n_rr, n_pAp of the device control block
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h.stop, out[1] = h.stop_next, out[2] = h.iters, out[3] = h.n_rr, out[4] = h.n_pAp;
}

static bool spmv_can_fuse_dot(const sb_cg* s)
{ // p.Ap partials in the SpMV epilogue: the wave-per-chunk kernels (SCS C=64, or CRS through its mirror)
  return s->fused && (s->A->fmt == 1 ? s->A->C == 64 : spmv_uses_patterns(s->A));
}

// levels 1-2 of the reduction + the scalar step: one 1-workgroup launch after the producer
// (several ranks: local sum -> RCCL all-reduce in place on the stream -> scalar step;
// MPI_Allreduce of src/comm.c:659)
template <int MODE> static void scalar_launch(sb_cg* s, int defer_x = 0)
{
  if (multi_rank() && g.p2pOn) { // local reduce, in-kernel all-reduce and scalar step in ONE launch
    hipLaunchKernelGGL((cg_scalar_k<MODE, true>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, s->partials,
        s->S, s->rr_hist, s->pAp_hist, 0, defer_x, (const P2PView*)g.p2pView, ++g.p2pSeq);
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL((cg_scalar_k<MODE, true>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, s->partials,
      s->S, s->rr_hist, s->pAp_hist, multi_rank() ? 1 : 0, defer_x, (const P2PView*)nullptr, 0ull);
  HIP_CHECK(hipGetLastError());
  if (multi_rank()) {
    mark(s, R_DDOT);
    sb_comm_reduction(&s->S->local, 1);
    mark(s, R_COMM);
    hipLaunchKernelGGL((cg_scalar_k<MODE, false>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, s->partials,
        s->S, s->rr_hist, s->pAp_hist, 0, defer_x, (const P2PView*)nullptr, 0ull);
    HIP_CHECK(hipGetLastError());
  }
}

static void spmv_event(sb_cg* s)
{
  if (!s->spmvTiming) return;
  if (s->spmvEvUsed == s->spmvEv.size()) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    s->spmvEv.push_back(e);
  }
  HIP_CHECK(hipEventRecord(s->spmvEv[s->spmvEvUsed++], g.stream));
}

// one loop body of solveCG (src/CGSolver.c:108-128).  Fused path: the r.r partials of the
// NEXT body come out of this body's x/r update, and its beta + loop test are taken right
// after it, so a body is: p update | SpMV (+p.Ap partials) | alpha | x/r update (+r.r
// partials) | beta, loop test.
static void loop_body(sb_cg* s, int k)
{
  const uint32_t n = s->nr;
  const int* stop  = &s->S->stop;
  dim3 gridV(stream_grid(n / 2 + 1, 256)), blockV(256);
  if (k == 1) {
    if (n) hipLaunchKernelGGL(cg_update_p, gridV, blockV, 0, g.stream, n, s->r, s->p, (double*)nullptr, s->S, 1); // p = r (:109)
    mark(s, R_WAXPBY);
  } else {
    if (!s->fused) { // rtrans = r.r ; beta (:111-113)
      launch_dot_spans(0, n, s->r, s->r, nullptr, nullptr, s->S, s->partials, stop);
      scalar_launch<1>(s);
      mark(s, R_DDOT);
    }
    if (n) // p = r + beta p (:114); fused path: also the x update owed by the previous body (:127)
      hipLaunchKernelGGL(cg_update_p, gridV, blockV, 0, g.stream, n, s->r, s->p, s->fused ? s->x : (double*)nullptr, s->S, 0);
    mark(s, R_WAXPBY);
  }
  HIP_CHECK(hipGetLastError());
  static const bool overlapHalo = !(getenv("SB_HALO_OVERLAP") && atoi(getenv("SB_HALO_OVERLAP")) == 0);
  if (overlapHalo && multi_rank() && s->halo && spmv_can_fuse_dot(s) && spmv_can_split(s->A)) {
    // :122-126 with the halo exchange hidden behind the interior tiles: the exchange (pack,
    // send/recv into the tail of p) runs on a second stream while the tiles that touch no
    // halo column are multiplied; the halo-touching tiles follow.  RCCL calls on the one
    // communicator stay ordered: the exchange is complete (event) before anything later.
    if (g.hasXport) halo_exchange(s->halo, s->p, stop); // host-mediated: synchronous anyway
    else {
      HIP_CHECK(hipEventRecord(g.evFork, g.stream));
      HIP_CHECK(hipStreamWaitEvent(g.stream2, g.evFork, 0));
      halo_exchange(s->halo, s->p, stop, g.stream2);
      HIP_CHECK(hipEventRecord(g.evJoin, g.stream2));
    }
    mark(s, R_COMM);
    spmv_event(s);
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop, 1);
    if (!g.hasXport) HIP_CHECK(hipStreamWaitEvent(g.stream, g.evJoin, 0));
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop, 2);
    spmv_event(s);
    mark(s, R_SPMVM);
    goto alpha_step;
  }
  halo_exchange(s->halo, s->p, stop); // :122
  mark(s, R_COMM);
  spmv_event(s);
  if (spmv_can_fuse_dot(s)) { // Ap = A p, alpha = rtrans / p.Ap (:123-126)
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop);
    spmv_event(s);
    mark(s, R_SPMVM);
  } else {
    launch_spmv(s->A, s->p, s->Ap, nullptr, stop);
    spmv_event(s);
    mark(s, R_SPMVM);
    launch_dot_spans(0, n, s->p, s->Ap, nullptr, nullptr, s->S, s->partials, stop);
  }
alpha_step:
  scalar_launch<2>(s);
  mark(s, R_DDOT);
  if (s->fused) { // r -= alpha Ap (:128) + next r.r, beta, loop test; x += alpha p (:127) is owed
    launch_dot_spans(3, n, s->p, s->Ap, s->x, s->r, s->S, s->partials, stop);
    mark(s, R_WAXPBY);
    scalar_launch<1>(s, 1);
    mark(s, R_DDOT);
  } else if (n) {
    hipLaunchKernelGGL(waxpby_sdev_k, gridV, blockV, 0, g.stream, n, s->x, &s->S->alpha, s->p, s->x, stop);
    hipLaunchKernelGGL(waxpby_sdev_k, gridV, blockV, 0, g.stream, n, s->r, &s->S->neg_alpha, s->Ap, s->r, stop);
    HIP_CHECK(hipGetLastError());
    mark(s, R_WAXPBY);
  }
}

static void ensure_hist(sb_cg* s, int cap)
{
  if (cap <= s->hist_cap) return;
  sb_free(s->rr_hist), sb_free(s->pAp_hist);
  s->hist_cap = cap;
  s->rr_hist  = (double*)sb_malloc((size_t)cap * sizeof(double));
  s->pAp_hist = (double*)sb_malloc((size_t)cap * sizeof(double));
  drop_graph(s); // captured pointers are stale
}

static void run_body_maybe_graph(sb_cg* s, int k)
{ // k >= 2 bodies are iteration-invariant (k lives in the device control block)
  if (k < 2 || !s->use_graph || multi_rank() || s->timing || s->spmvTiming) {
    loop_body(s, k);
    return;
  }
  if (!s->graphReady) {
    hipGraph_t graph;
    HIP_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
    loop_body(s, 2);
    HIP_CHECK(hipStreamEndCapture(g.stream, &graph));
    HIP_CHECK(hipGraphInstantiate(&s->iterGraph, graph, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(graph));
    s->graphReady = true;
  }
  HIP_CHECK(hipGraphLaunch(s->iterGraph, g.stream));
}

void sb_cg_start(sb_cg* s, int itermax, double eps)
{
  need_init();
  const uint32_t n = s->nr;
  ensure_hist(s, itermax + 2);
  s->timing  = !s->fused; // the reference-shaped op list is the one that gets the region table
  s->evUsed  = 0;
  memset(&s->hostS, 0, sizeof s->hostS);
  s->hostS.itermax  = itermax;
  s->hostS.eps      = eps;
  s->hostS.hist_cap = s->hist_cap;
  HIP_CHECK(hipMemcpyAsync(s->S, &s->hostS, sizeof(CgScalars), hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipMemsetAsync(s->x, 0, (size_t)n * sizeof(double), g.stream)); // x0 = 0 (:28)
  HIP_CHECK(hipMemsetAsync(s->p, 0, (size_t)s->nc * sizeof(double), g.stream));
  mark(s, -1);
  // prologue, src/CGSolver.c:94-100
  launch_waxpby(n, 1.0, s->x, 0.0, s->x, s->p, nullptr);
  mark(s, R_WAXPBY);
  halo_exchange(s->halo, s->p, nullptr);
  mark(s, R_COMM);
  launch_spmv(s->A, s->p, s->Ap, nullptr, nullptr);
  mark(s, R_SPMVM);
  if (s->fused) {
    launch_dot_spans(2, n, s->b, s->Ap, nullptr, s->r, s->S, s->partials, nullptr);
    mark(s, R_WAXPBY);
  } else {
    launch_waxpby(n, 1.0, s->b, -1.0, s->Ap, s->r, nullptr);
    mark(s, R_WAXPBY);
    launch_dot_spans(0, n, s->r, s->r, nullptr, nullptr, s->S, s->partials, nullptr);
  }
  scalar_launch<0>(s);
  mark(s, R_DDOT);
  s->k_next  = 1;
  s->started = true;
}

void sb_cg_run_iters(sb_cg* s, int iters)
{
  need_init();
  if (!s->started) SB_FATAL("sb_cg_run_iters before sb_cg_start");
  for (int i = 0; i < iters; i++) run_body_maybe_graph(s, s->k_next++);
}

int sb_cg_finish(sb_cg* s)
{
  need_init();
  if (s->nr) { // the x update the last body left to "the next p update": nobody comes after it
    hipLaunchKernelGGL(cg_x_finalize, dim3(stream_grid(s->nr, 256)), dim3(256), 0, g.stream, s->nr, s->x, s->p, s->S);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(&s->S->x_pending, 0, sizeof(int), g.stream));
  }
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
  if (h.p2p_error)
    SB_FATAL("rank %d: a peer's contribution to an in-kernel all-reduce did not arrive within 2 s "
             "(SB_P2P=0 selects the RCCL all-reduce)", g.rank);
  if (s->timing) {
    for (double& v : s->region_ms) v = 0.0;
    for (size_t i = 1; i < s->evUsed; i++) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, s->ev[i - 1], s->ev[i]));
      if (s->evRegion[i] >= 0) s->region_ms[s->evRegion[i]] += ms;
    }
  }
  s->timing = false;
  return h.iters + 1; // the value of k when the reference's for loop exits (:107,:140)
}

int sb_cg_solve(sb_cg* s, int itermax, double eps)
{
  sb_cg_start(s, itermax, eps);
  HIP_CHECK(hipEventRecord(s->evLoop0, g.stream));
  sb_cg_run_iters(s, itermax > 1 ? itermax - 1 : 0);
  HIP_CHECK(hipEventRecord(s->evLoop1, g.stream));
  const int k = sb_cg_finish(s);
  HIP_CHECK(hipEventElapsedTime(&s->loop_ms, s->evLoop0, s->evLoop1));
  return k;
}

int sb_cg_history(const sb_cg* s, double* rr_out, int rr_cap, double* pAp_out, int pAp_cap, int* n_pAp)
{
  need_init();
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
  int nrr = h.n_rr < s->hist_cap ? h.n_rr : s->hist_cap;
  int npa = h.n_pAp < s->hist_cap ? h.n_pAp : s->hist_cap;
  if (nrr > rr_cap) nrr = rr_cap;
  if (npa > pAp_cap) npa = pAp_cap;
  if (nrr > 0) HIP_CHECK(hipMemcpy(rr_out, s->rr_hist, (size_t)nrr * sizeof(double), hipMemcpyDeviceToHost));
  if (npa > 0) HIP_CHECK(hipMemcpy(pAp_out, s->pAp_hist, (size_t)npa * sizeof(double), hipMemcpyDeviceToHost));
  if (n_pAp) *n_pAp = npa;
  return nrr;
}

void sb_cg_solution(const sb_cg* s, double* x_host)
{
  need_init();
  double* tmp = scratch_ws(1, s->nr);
  sb_unpermute(s->A, s->x, tmp);
  sb_d2h(x_host, tmp, (size_t)s->nr * sizeof(double));
}

double sb_cg_check_residual(const sb_cg* s)
{
  need_init();
  if (!s->xexact || s->nr == 0) return 0.0;
  const uint32_t blocks = stream_grid(s->nr, 256);
  double* q             = scratch_partials(blocks);
  hipLaunchKernelGGL(max_abs_diff_partials, dim3(blocks), dim3(256), 0, g.stream, s->nr, s->x, s->xexact, q);
  HIP_CHECK(hipGetLastError());
  std::vector<double> h(blocks);
  sb_d2h(h.data(), q, blocks * sizeof(double));
  double m = 0.0;
  for (double v : h)
    if (v > m) m = v;
  if (multi_rank()) { // commReduction(&residual, MAX), src/CGSolver.c:55
    sb_h2d(g.scalar, &m, sizeof m);
    sb_comm_reduction(g.scalar, 0);
    sb_d2h(&m, g.scalar, sizeof m);
  }
  return m;
}

double sb_debug_stream_read_gbs(size_t bytes, int reps)
{ // raw read ceiling of this device: `reps` passes over a `bytes`-sized buffer
  need_init();
  double2* buf = nullptr;
  HIP_CHECK(hipMalloc(&buf, bytes));
  HIP_CHECK(hipMemsetAsync(buf, 0, bytes, g.stream));
  const size_t n2 = bytes / sizeof(double2);
  dim3 grid((unsigned)g.prop.multiProcessorCount * 8), block(256);
  hipLaunchKernelGGL(stream_read_k, grid, block, 0, g.stream, buf, n2, g.scalar);
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(a, g.stream));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(stream_read_k, grid, block, 0, g.stream, buf, n2, g.scalar);
  HIP_CHECK(hipEventRecord(b, g.stream));
  HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  HIP_CHECK(hipEventDestroy(a));
  HIP_CHECK(hipEventDestroy(b));
  HIP_CHECK(hipFree(buf));
  return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

double sb_cg_loop_ms(const sb_cg* s) { return (double)s->loop_ms; }

void sb_cg_region_ms(const sb_cg* s, double out[4])
{
  for (int i = 0; i < 4; i++) out[i] = s->region_ms[i];
}
