/* sbh_comm.c -- rank bootstrap, 1-D block-row partition with halo plan, and the
 * run-time halo / reduction calls, for one rank per GPU.
 *
 * What it replaces in the reference: commInit/commFinalize (src/comm.c:863-905),
 * commDistributeMatrix (:311-412), commPartition (:414-625) with its helpers
 * buildIndexMapping (:40-114) and buildElementsToSend (:116-182), commExchange
 * (:627-651), commReduction (:653-662).  The reference talks MPI everywhere; here
 * setup traffic goes through the launcher-provided sbh_exchange (RCCL by default),
 * and the per-iteration traffic is RCCL over xGMI inside the HIP layer.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <unistd.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

static const sbh_exchange* g_xchg = NULL;

void commSetExchange(const sbh_exchange* x) { g_xchg = x; }

/* ---- setup exchange over RCCL --------------------------------------------------- */
static void rccl_allgather(void* ctx, const int* mine, int n, int* all)
{
  (void)ctx;
  sb_comm_allgather_bytes(mine, n * (int)sizeof(int), all);
}
static void rccl_alltoallv(void* ctx, const int* sendbuf, const int* sendcounts, const int* sdispls,
    int* recvbuf, const int* recvcounts, const int* rdispls)
{
  (void)ctx;
  sb_comm_alltoallv_ints(sendbuf, sendcounts, sdispls, recvbuf, recvcounts, rdispls);
}
static const sbh_exchange g_rccl_exchange = { NULL, rccl_allgather, rccl_alltoallv };
const sbh_exchange* sbh_exchange_rccl(void) { return &g_rccl_exchange; }

/* ---- bootstrap --------------------------------------------------------------------- */
static int env_int(const char* const* names, int dflt)
{
  for (; *names; names++) {
    const char* v = getenv(*names);
    if (v && *v) return atoi(v);
  }
  return dflt;
}

/* One process per GPU.  Rank and size come from the launcher's environment
 * (torchrun: RANK/WORLD_SIZE/LOCAL_RANK; mpiexec.hydra used as a plain process
 * launcher: PMI_RANK/PMI_SIZE; or SB_RANK/SB_SIZE).  The RCCL unique id travels
 * through a file in /tmp that rank 0 publishes atomically. */
void commInit(Comm* c, int argc, char** argv)
{
  (void)argc, (void)argv;
  static const char* const rankNames[]  = { "SB_RANK", "RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", NULL };
  static const char* const sizeNames[]  = { "SB_SIZE", "WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", NULL };
  static const char* const localNames[] = { "SB_LOCAL_RANK", "LOCAL_RANK", "MPI_LOCALRANKID",
    "OMPI_COMM_WORLD_LOCAL_RANK", NULL };
  memset(c, 0, sizeof *c);
  c->rank    = env_int(rankNames, 0);
  c->size    = env_int(sizeNames, 1);
  c->logFile = NULL;
  int ndev   = sb_device_count();
  if (ndev <= 0) {
    fprintf(stderr, "sparsebench: no HIP device visible; there is no CPU fallback\n");
    exit(EXIT_FAILURE);
  }
  int local = env_int(localNames, c->rank % ndev);
  sb_init(local % ndev);
  /* more ranks than devices on this node: ranks share GPUs (a rehearsal).  The one-launch vector phase of the
   * CG loop needs its grid resident all at once and must not be used then (include/sbhip.h: sb_cg_set_fused) */
  if (c->size > ndev) setenv("SB_SHARED_GPU", "1", 0);
  if (c->size > 1) {
    unsigned char id[SB_UNIQUE_ID_BYTES];
    char path[512], tmp[544];
    const char* f = getenv("SB_ID_FILE");
    const char* port = getenv("MASTER_PORT");
    if (f) snprintf(path, sizeof path, "%s", f);
    else snprintf(path, sizeof path, "/tmp/sbhip_id_%s_%d", port ? port : "0", (int)getppid());
    if (c->rank == 0) {
      sb_comm_unique_id(id);
      snprintf(tmp, sizeof tmp, "%s.tmp", path);
      FILE* o = fopen(tmp, "wb");
      if (!o || fwrite(id, 1, sizeof id, o) != sizeof id) {
        fprintf(stderr, "sparsebench: cannot write %s\n", tmp);
        exit(EXIT_FAILURE);
      }
      fclose(o);
      rename(tmp, path);
    } else {
      FILE* i = NULL;
      for (int tries = 0; tries < 6000 && !(i = fopen(path, "rb")); tries++) usleep(10000);
      if (!i || fread(id, 1, sizeof id, i) != sizeof id) {
        fprintf(stderr, "sparsebench: rank %d could not read the RCCL id from %s\n", c->rank, path);
        exit(EXIT_FAILURE);
      }
      fclose(i);
    }
    sb_comm_init(c->rank, c->size, id);
    sb_comm_barrier();
    if (c->rank == 0) unlink(path);
    commSetExchange(sbh_exchange_rccl());
  }
}

void commFinalize(Comm* c)
{
  if (c->dev) sb_halo_free((sb_halo*)c->dev);
  c->dev = NULL;
  free(c->elementsToSend), free(c->sources), free(c->recvCounts), free(c->rdispls);
  free(c->destinations), free(c->sendCounts), free(c->sdispls), free(c->externalGlobal);
  c->elementsToSend = c->sources = c->recvCounts = c->rdispls = NULL;
  c->destinations = c->sendCounts = c->sdispls = NULL;
  c->externalGlobal = NULL;
  if (getenv("SB_COPY_REPORT") && sb_is_initialized()) {
    /* what crossed PCIe through this layer during the run, and what the allocation hook handed out (tests/test_gpu_dropin.py) */
    uint64_t n[4];
    sb_copy_counters(n);
    fprintf(stderr, "sbhip copies: h2d %llu calls %llu bytes, d2h %llu calls %llu bytes; allocate(): last kind %d (%s)\n",
        (unsigned long long)n[0], (unsigned long long)n[1], (unsigned long long)n[2], (unsigned long long)n[3],
        sbh_allocate_kind(), sb_host_visible_reason());
  }
  sb_finalize();
}

void commBarrier(void)
{
  if (sb_is_initialized()) sb_comm_barrier();
}

void commAbort(Comm* c, char* msg)
{ /* src/comm.c:880-891: message from the master, then a clean exit */
  if (commIsMaster(c)) printf("%s\n", msg);
  if (sb_is_initialized()) sb_finalize();
  exit(EXIT_SUCCESS);
}

/* src/comm.c:185-250: banner; the format name comes from the per-format library (dropin.c), as the reference's
 * comes from its -DCRS|-DSCS build */
void sbh_print_banner(Comm* c, const char* fmt)
{
  if (!commIsMaster(c)) return;
  printf(HLINE);
  printf("SparseBench CG hot path -- MI355X HIP build (%s)\n", sb_version());
  printf("Device: %s, %d CUs\n", sb_device_name(), sb_num_cus());
  if (fmt) printf("Using %s matrix format, %s precision floats and integer type %s\n", fmt, PRECISION_STRING, UINT_STRING);
  else printf("Using %s precision floats and integer type %s\n", PRECISION_STRING, UINT_STRING);
  if (c->size > 1) printf("RCCL parallel using %d ranks (one per GPU)\n", c->size);
  else printf("Running with only one process!\n");
  printf(HLINE);
}

void commPrintConfig(Comm* c, CG_UINT nr, CG_UINT nnz, CG_UINT startRow, CG_UINT stopRow)
{
  printf("Rank %d of %d: %u rows (%u..%u), %u nonzeros, %d externals, %d to send, "
         "%d in-neighbours, %d out-neighbours\n",
      c->rank, c->size, nr, startRow, stopRow, nnz, c->externalCount, c->totalSendCount, c->indegree,
      c->outdegree);
}

/* VERBOSE-build diagnostics (src/comm.c:664-861 print the same content into out-<rank>.txt) */
void commGMatrixDump(Comm* c, GMatrix* m)
{
  FILE* f = c->logFile ? c->logFile : stdout;
  fprintf(f, "Matrix: %u total non zeroes, total number of rows %u\n", m->totalNnz, m->totalNr);
  fprintf(f, "Matrix: %u local non zeroes, local number of rows %u (%u..%u)\n", m->rowPtr ? m->rowPtr[m->nr] : 0u,
      m->nr, m->startRow, m->stopRow);
  for (CG_UINT i = 0; i < m->nr; i++) {
    fprintf(f, "Row [%u]: ", i);
    for (CG_UINT j = m->rowPtr[i]; j < m->rowPtr[i + 1]; j++) fprintf(f, "[%u]:%.2f ", m->entries[j].col, m->entries[j].val);
    fprintf(f, "\n");
  }
  fflush(f);
}

void commVectorDump(Comm* c, CG_FLOAT* v, CG_UINT size, char* name)
{
  FILE* f      = c->logFile ? c->logFile : stdout;
  CG_FLOAT* h  = v;
  if (sb_is_initialized() && sb_is_device_ptr(v)) {
    h = (CG_FLOAT*)malloc(((size_t)size + 1) * sizeof(CG_FLOAT));
    sb_sync();
    sb_d2h(h, v, (size_t)size * sizeof(CG_FLOAT));
  }
  fprintf(f, "Vector %s Rank %d\n", name, c->rank);
  for (CG_UINT i = 0; i < size; i++) fprintf(f, "element[%u] %f\n", i, h[i]);
  fflush(f);
  if (h != v) free(h);
}

/* ---- row split of file matrices: src/comm.c:35-38, :347-361 -------------------------- */
static void rows_of_rank(int rank, int size, int N, int* first, int* last)
{
  int base = N / size, extra = N % size;
  *first = rank * base + (rank < extra ? rank : extra);
  *last  = *first + base + (rank < extra ? 1 : 0) - 1;
}

/* Every rank holds the whole file (our own driver: each rank calls MMMatrixRead) -- a rank
 * keeps the entries of its own row range.  One rank: alias, as src/comm.c:404-411. */
void sbh_distribute_local(Comm* c, MMMatrix* m, MMMatrix* mLocal)
{
  int first, last;
  rows_of_rank(c->rank, c->size, m->nr, &first, &last);
  if (commIsMaster(c) && c->size > 1)
    for (int r = 0; r < c->size; r++) {
      int a, b;
      rows_of_rank(r, c->size, m->nr, &a, &b);
      printf("Rank %d start %d stop %d\n", r, a, b);
    }
  size_t lo = 0, hi = m->count;
  if (c->size > 1) {
    while (lo < m->count && m->entries[lo].row < first) lo++;
    hi = lo;
    while (hi < m->count && m->entries[hi].row <= last) hi++;
  }
  mLocal->entries  = m->entries + lo;
  mLocal->count    = hi - lo;
  mLocal->nnz      = (int)(hi - lo);
  mLocal->startRow = first;
  mLocal->stopRow  = last;
  mLocal->nr       = last - first + 1;
  mLocal->totalNr  = m->nr;
  mLocal->totalNnz = m->nnz;
}

/* The reference's contract (src/comm.c:311-402, src/main.c:63-70): ONLY the master has read the
 * file; the other ranks pass an uninitialised MMMatrix.  So with several ranks the master's
 * header and entries are authoritative.  As in the reference (MPI_Bcast of the counts +
 * MPI_Scatterv of the entries, src/comm.c:340-383) every rank receives ONLY the entries of its own
 * row range: the master announces the per-rank counts in one small all-gather and then sends each
 * rank its slice through the setup exchange's all-to-all, in bounded rounds (round 2 broadcast the
 * whole matrix to everybody through an all-gather: P times the traffic and P x 16 MiB per rank).
 * Deviation from the reference, on purpose: startRow / stopRow are the rank's ROW RANGE (the split
 * of src/comm.c:35-38), not the row numbers of its first / last entry (src/comm.c:385-387) -- the
 * two differ only when a range starts or ends with empty rows, where the reference would shift
 * the ownership of those rows (and fault on a rank without entries); a rank's vectors here always
 * cover its whole range. */
void commDistributeMatrix(Comm* c, MMMatrix* m, MMMatrix* mLocal)
{
  if (c->size == 1) {
    sbh_distribute_local(c, m, mLocal);
    return;
  }
  if (!g_xchg) {
    fprintf(stderr, "commDistributeMatrix: %d ranks but no setup exchange (commSetExchange)\n", c->size);
    exit(EXIT_FAILURE);
  }
  const int P = c->size;
  /* header from the master: nr, nnz, then the number of entries of every rank's row range */
  const int H = 2 + P;
  int* hdr    = (int*)calloc((size_t)H, sizeof(int));
  size_t* lo  = (size_t*)calloc((size_t)P + 1, sizeof(size_t));
  if (commIsMaster(c)) {
    hdr[0] = m->nr, hdr[1] = m->nnz;
    size_t at = 0;
    for (int r = 0; r < P; r++) { /* entries are sorted by row (MMMatrixRead) */
      int a, b;
      rows_of_rank(r, P, m->nr, &a, &b);
      while (at < m->count && m->entries[at].row < a) at++;
      lo[r] = at;
      while (at < m->count && m->entries[at].row <= b) at++;
      if (at - lo[r] > 0x7FFFFFFFu / 4u) {
        fprintf(stderr, "commDistributeMatrix: rank %d's slice exceeds the setup exchange's 32-bit counts\n", r);
        exit(EXIT_FAILURE);
      }
      hdr[2 + r] = (int)(at - lo[r]);
    }
    lo[P] = at;
  }
  int* allHdr = (int*)malloc((size_t)P * H * sizeof(int));
  g_xchg->allgather_ints(g_xchg->ctx, hdr, H, allHdr);
  const int totalNr = allHdr[0], totalNnz = allHdr[1]; /* rank 0's slice comes first */
  const size_t mineCount = (size_t)allHdr[2 + c->rank];
  size_t maxCount = 0;
  for (int r = 0; r < P; r++)
    if ((size_t)allHdr[2 + r] > maxCount) maxCount = (size_t)allHdr[2 + r];
  free(allHdr), free(hdr);
  int first, last;
  rows_of_rank(c->rank, P, totalNr, &first, &last);
  if (commIsMaster(c))
    for (int r = 0; r < P; r++) {
      int a, b;
      rows_of_rank(r, P, totalNr, &a, &b);
      printf("Rank %d start %d stop %d\n", r, a, b);
    }
  const size_t PIECE = 1u << 20; /* entries per rank and round: 16 MiB */
  MMEntry* keep = (MMEntry*)malloc((mineCount ? mineCount : 1) * sizeof(MMEntry));
  int *scnt = (int*)calloc((size_t)P, sizeof(int)), *sdsp = (int*)calloc((size_t)P, sizeof(int));
  int *rcnt = (int*)calloc((size_t)P, sizeof(int)), *rdsp = (int*)calloc((size_t)P, sizeof(int));
  int dummy[4] = { 0, 0, 0, 0 };
  /* the master packs a round's pieces back to back (transports stage the send buffer up to its last displacement) */
  const size_t roundCap = maxCount < PIECE ? maxCount : PIECE;
  MMEntry* stage = commIsMaster(c) ? (MMEntry*)malloc(((size_t)P * roundCap + 1) * sizeof(MMEntry)) : NULL;
  for (size_t at = 0; at < maxCount; at += PIECE) {
    /* this round: entries [at, at + PIECE) of every rank's slice */
    const size_t mineN = at < mineCount ? (mineCount - at < PIECE ? mineCount - at : PIECE) : 0;
    if (commIsMaster(c)) {
      size_t fill = 0;
      for (int r = 1; r < P; r++) {
        const size_t have = lo[r + 1] - lo[r];
        const size_t n    = at < have ? (have - at < PIECE ? have - at : PIECE) : 0;
        if (n) memcpy(stage + fill, m->entries + lo[r] + at, n * sizeof(MMEntry));
        scnt[r] = (int)(n * 4), sdsp[r] = (int)(fill * 4);
        fill += n;
      }
      if (mineN) memcpy(keep + at, m->entries + lo[0] + at, mineN * sizeof(MMEntry)); /* its own slice: no transfer */
    } else {
      rcnt[0] = (int)(mineN * 4), rdsp[0] = 0;
    }
    g_xchg->alltoallv_ints(g_xchg->ctx, commIsMaster(c) ? (const int*)stage : dummy, scnt, sdsp,
        commIsMaster(c) || !mineN ? dummy : (int*)(keep + at), rcnt, rdsp);
  }
  free(stage), free(scnt), free(sdsp), free(rcnt), free(rdsp), free(lo);
  mLocal->entries  = keep;
  mLocal->count    = mineCount;
  mLocal->nnz      = (int)mineCount;
  mLocal->startRow = first;
  mLocal->stopRow  = last;
  mLocal->nr       = last - first + 1;
  mLocal->totalNr  = totalNr;
  mLocal->totalNnz = totalNnz;
}

/* ---- partition + halo plan --------------------------------------------------------- */
typedef struct {
  CG_UINT* keys;
  int* vals;
  size_t cap, used;
} idmap;

static size_t slot_of(const idmap* h, CG_UINT k)
{
  size_t i = ((size_t)k * 0x9E3779B1u) & (h->cap - 1);
  while (h->vals[i] >= 0 && h->keys[i] != k) i = (i + 1) & (h->cap - 1);
  return i;
}

static void idmap_init(idmap* h, size_t cap)
{
  h->cap  = cap;
  h->used = 0;
  h->keys = (CG_UINT*)malloc(cap * sizeof(CG_UINT));
  h->vals = (int*)malloc(cap * sizeof(int));
  for (size_t i = 0; i < cap; i++) h->vals[i] = -1;
}

static void idmap_grow(idmap* h)
{
  idmap n;
  idmap_init(&n, h->cap * 4);
  for (size_t i = 0; i < h->cap; i++)
    if (h->vals[i] >= 0) {
      size_t s  = slot_of(&n, h->keys[i]);
      n.keys[s] = h->keys[i], n.vals[s] = h->vals[i];
    }
  n.used = h->used;
  free(h->keys), free(h->vals);
  *h = n;
}

void commPartition(Comm* c, GMatrix* A)
{
  c->externalCount = c->totalSendCount = c->indegree = c->outdegree = 0;
  if (c->size == 1) return; /* serial build of the reference: no-op (src/comm.c:416,624) */
  if (!g_xchg) {
    fprintf(stderr, "commPartition: %d ranks but no setup exchange installed (commSetExchange)\n", c->size);
    exit(EXIT_FAILURE);
  }
  const int P      = c->size;
  const CG_UINT nr = A->nr, first = A->startRow, last = A->stopRow;

  /* 1. external columns in first-seen order (src/comm.c:452-473; hash instead of BST) */
  idmap seen;
  idmap_init(&seen, 1u << 12);
  size_t extCap  = 1024;
  CG_UINT* extId = (CG_UINT*)malloc(extCap * sizeof(CG_UINT));
  int nExt       = 0;
  for (CG_UINT i = 0; i < nr; i++)
    for (CG_UINT j = A->rowPtr[i]; j < A->rowPtr[i + 1]; j++) {
      const CG_UINT g = A->entries[j].col;
      if (g >= first && g <= last) continue;
      size_t s = slot_of(&seen, g);
      if (seen.vals[s] >= 0) continue;
      seen.keys[s] = g, seen.vals[s] = nExt, seen.used++;
      if ((size_t)nExt == extCap) extId = (CG_UINT*)realloc(extId, (extCap *= 2) * sizeof(CG_UINT));
      extId[nExt++] = g;
      if (seen.used * 2 > seen.cap) idmap_grow(&seen);
    }

  /* 2. owners (src/comm.c:496-520): last rank whose first row <= id */
  int* starts = (int*)malloc((size_t)P * sizeof(int));
  int mine    = (int)first;
  g_xchg->allgather_ints(g_xchg->ctx, &mine, 1, starts);
  int* owner = (int*)malloc(((size_t)nExt + 1) * sizeof(int));
  int* want  = (int*)calloc((size_t)P, sizeof(int)); /* want[s]: how many ids I need from s */
  for (int i = 0; i < nExt; i++) {
    int lo = 0, hi = P - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) / 2;
      if ((CG_UINT)starts[mid] <= extId[i]) lo = mid;
      else hi = mid - 1;
    }
    owner[i] = lo;
    want[lo]++;
  }

  /* 3. local numbering (src/comm.c:60-110): externals follow the nr local columns,
   * grouped by owner -- ascending owner, which is the order the receive displacements
   * assume -- and first-seen inside a group */
  int* base = (int*)malloc(((size_t)P + 1) * sizeof(int));
  base[0]   = 0;
  for (int s = 0; s < P; s++) base[s + 1] = base[s] + want[s];
  int* fill         = (int*)calloc((size_t)P, sizeof(int));
  int* localOf      = (int*)malloc(((size_t)nExt + 1) * sizeof(int));
  c->externalGlobal = (CG_UINT*)malloc(((size_t)nExt + 1) * sizeof(CG_UINT));
  for (int i = 0; i < nExt; i++) {
    int pos                = base[owner[i]] + fill[owner[i]]++;
    localOf[i]             = (int)nr + pos;
    c->externalGlobal[pos] = extId[i];
  }
  for (CG_UINT i = 0; i < nr; i++)
    for (CG_UINT j = A->rowPtr[i]; j < A->rowPtr[i + 1]; j++) {
      const CG_UINT g = A->entries[j].col;
      if (g >= first && g <= last) A->entries[j].col = g - first;
      else A->entries[j].col = (CG_UINT)localOf[seen.vals[slot_of(&seen, g)]];
    }
  A->nc            = A->nc + (CG_UINT)nExt; /* src/comm.c:616 */
  c->externalCount = nExt;
  sb_set_external_ids(c->externalGlobal, (uint32_t)nExt); /* window layout hint for the upload that follows (sbhip.h) */

  c->sources    = (int*)malloc(((size_t)P + 1) * sizeof(int));
  c->recvCounts = (int*)malloc(((size_t)P + 1) * sizeof(int));
  c->rdispls    = (int*)malloc(((size_t)P + 1) * sizeof(int));
  for (int s = 0; s < P; s++)
    if (want[s] > 0) {
      c->sources[c->indegree]    = s;
      c->recvCounts[c->indegree] = want[s];
      c->rdispls[c->indegree]    = base[s];
      c->indegree++;
    }

  /* 4. tell every owner which of its rows I need (src/comm.c:116-166): the counts
   * matrix by all-gather, the id lists by all-to-all */
  int* wantAll = (int*)malloc((size_t)P * P * sizeof(int)); /* wantAll[d*P+s] */
  g_xchg->allgather_ints(g_xchg->ctx, want, P, wantAll);
  int* give     = (int*)malloc((size_t)P * sizeof(int)); /* give[d]: ids rank d needs from me */
  int* giveDisp = (int*)malloc(((size_t)P + 1) * sizeof(int));
  giveDisp[0]   = 0;
  for (int d = 0; d < P; d++) {
    give[d]         = wantAll[(size_t)d * P + c->rank];
    giveDisp[d + 1] = giveDisp[d] + give[d];
  }
  c->totalSendCount = giveDisp[P];
  c->elementsToSend = (int*)malloc(((size_t)c->totalSendCount + 1) * sizeof(int));
  g_xchg->alltoallv_ints(g_xchg->ctx, (const int*)c->externalGlobal, want, base, c->elementsToSend,
      give, giveDisp);
  for (int i = 0; i < c->totalSendCount; i++) c->elementsToSend[i] -= (int)first;

  c->destinations = (int*)malloc(((size_t)P + 1) * sizeof(int));
  c->sendCounts   = (int*)malloc(((size_t)P + 1) * sizeof(int));
  c->sdispls      = (int*)malloc(((size_t)P + 1) * sizeof(int));
  for (int d = 0; d < P; d++)
    if (give[d] > 0) {
      c->destinations[c->outdegree] = d;
      c->sendCounts[c->outdegree]   = give[d];
      c->sdispls[c->outdegree]      = giveDisp[d];
      c->outdegree++;
    }

  free(seen.keys), free(seen.vals), free(extId), free(starts), free(owner), free(want);
  free(base), free(fill), free(localOf), free(wantAll), free(give), free(giveDisp);
}

/* ---- run time ------------------------------------------------------------------------- */
void sbh_comm_attach_halo(Comm* c, CG_UINT nr, const CG_UINT* oldToNewPerm)
{
  if (c->dev) sb_halo_free((sb_halo*)c->dev);
  c->dev = NULL;
  if (c->size == 1) return;
  c->dev = sb_halo_create(nr, c->outdegree, c->destinations, c->sendCounts, c->sdispls, c->indegree,
      c->sources, c->recvCounts, c->rdispls, c->elementsToSend, c->totalSendCount, c->externalCount,
      oldToNewPerm);
}

/* x: device vector with nc = numRows + externalCount entries */
void commExchange(Comm* c, CG_UINT numRows, CG_FLOAT* x)
{
  if (c->size == 1) return;
  if (!c->dev) sbh_comm_attach_halo(c, numRows, NULL);
  if (!sb_is_device_ptr(x)) {
    fprintf(stderr, "commExchange: x must live in HBM (sb_malloc) when running on %d ranks\n", c->size);
    exit(EXIT_FAILURE);
  }
  sb_halo_exchange((sb_halo*)c->dev, x);
}

void commReduction(CG_FLOAT* v, int op)
{
  if (!sb_is_initialized() || sb_comm_size() == 1) return;
  if (sb_is_device_ptr(v)) {
    sb_comm_reduction(v, op);
    return;
  }
  double* d = (double*)sb_malloc(sizeof(double));
  sb_h2d(d, v, sizeof(double));
  sb_comm_reduction(d, op);
  sb_d2h(v, d, sizeof(double));
  sb_free(d);
}
