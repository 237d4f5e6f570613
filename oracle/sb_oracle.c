/* sb_oracle.c -- CPU oracle for the SparseBench CG hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sb_oracle.h).  Plain C11, strict IEEE:
 * build with -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 *
 * Each function names the reference file:line whose algorithm it restates.
 * Nothing here is copied from the reference; data structures are SoA and the
 * multi-rank path runs P ranks lock-step inside one process.
 */
#define _GNU_SOURCE
#include "sb_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void* xmalloc(size_t n)
{
  void* p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n);
    exit(EXIT_FAILURE);
  }
  return p;
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ===================================================================== */
/* Generator: src/matrix.c:30-121 (matrixGenerate).                      */
/* One nx*ny*nz brick per rank stacked in z; 27-pt (or 7-pt) stencil;    */
/* diag 27, off-diag -1; x/y clipped by index, z by 0<=col<totalRows;    */
/* within-row order sz,sy,sx ascending; GLOBAL column ids.               */
/* ===================================================================== */
orc_gmatrix* orc_generate(int nx, int ny, int nz, int rank, int size, int use7pt)
{
  orc_gmatrix* g = (orc_gmatrix*)xmalloc(sizeof *g);
  uint32_t lnr   = (uint32_t)nx * (uint32_t)ny * (uint32_t)nz;
  uint32_t tnr   = lnr * (uint32_t)size;
  long start     = (long)lnr * rank;
  g->rowPtr      = (uint32_t*)xmalloc(((size_t)lnr + 1) * sizeof(uint32_t));
  g->col         = (uint32_t*)xmalloc((size_t)27 * lnr * sizeof(uint32_t));
  g->val         = (double*)xmalloc((size_t)27 * lnr * sizeof(double));
  size_t cur     = 0;
  uint32_t row   = 0;
  g->rowPtr[0]   = 0;
  for (int iz = 0; iz < nz; iz++)
    for (int iy = 0; iy < ny; iy++)
      for (int ix = 0; ix < nx; ix++) {
        long me = start + (long)iz * nx * ny + (long)iy * nx + ix;
        for (int sz = -1; sz <= 1; sz++)
          for (int sy = -1; sy <= 1; sy++)
            for (int sx = -1; sx <= 1; sx++) {
              long c = me + (long)sz * nx * ny + (long)sy * nx + sx;
              int inx = ix + sx >= 0 && ix + sx < nx;
              int iny = iy + sy >= 0 && iy + sy < ny;
              if (!(inx && iny && c >= 0 && c < (long)tnr)) continue;
              if (use7pt && sz * sz + sy * sy + sx * sx > 1) continue;
              g->val[cur] = (c == me) ? 27.0 : -1.0;
              g->col[cur] = (uint32_t)c;
              cur++;
            }
        g->rowPtr[++row] = (uint32_t)cur;
      }
  g->nr = g->nc = lnr;
  g->nnz        = 27u * lnr; /* the reference's upper bound, :35,:120 */
  g->nnzTrue    = (uint32_t)cur;
  g->totalNr    = tnr;
  g->totalNnz   = 27u * tnr; /* :38,:117 */
  g->startRow   = (uint32_t)start;
  g->stopRow    = (uint32_t)(start + lnr - 1);
  g->generated  = 1;
  return g;
}

/* ===================================================================== */
/* Matrix Market: src/matrix.c:123-229 (MMMatrixRead) + :231-269         */
/* (matrixConvertfromMM); header parsing restates what the call sites     */
/* use of NIST mmio (src/mmio.c): banner, skip %-comments, size line.     */
/* Entries: 1-based -> 0-based, symmetric files mirror off-diagonals,     */
/* then sort by column and stable-sort by row == stable sort by (row,col).*/
/* ===================================================================== */
typedef struct {
  int row, col;
  double val;
} mm_entry;

static void mm_merge_sort(mm_entry* a, mm_entry* tmp, size_t n)
{
  if (n < 2) return;
  size_t h = n / 2;
  mm_merge_sort(a, tmp, h);
  mm_merge_sort(a + h, tmp, n - h);
  size_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    int take_right = (a[j].row < a[i].row) ||
                     (a[j].row == a[i].row && a[j].col < a[i].col);
    tmp[k++] = take_right ? a[j++] : a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, n * sizeof *a);
}

static mm_entry* mm_read_all(const char* path, int* nrOut, size_t* countOut)
{
  FILE* f = fopen(path, "r");
  if (!f) {
    printf("Unable to open file.\n");
    exit(EXIT_FAILURE);
  }
  char line[4100], banner[64], obj[64], fmt[64], field[64], sym[64];
  if (!fgets(line, sizeof line, f) ||
      sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, sym) != 5 ||
      strcmp(banner, "%%MatrixMarket") != 0) {
    printf("Could not process Matrix Market banner.\n");
    exit(EXIT_FAILURE);
  }
  for (char* p = field; *p; p++) *p = (char)((*p >= 'A' && *p <= 'Z') ? *p + 32 : *p);
  for (char* p = sym; *p; p++) *p = (char)((*p >= 'A' && *p <= 'Z') ? *p + 32 : *p);
  for (char* p = fmt; *p; p++) *p = (char)((*p >= 'A' && *p <= 'Z') ? *p + 32 : *p);
  int is_pattern = strcmp(field, "pattern") == 0;
  int is_real    = strcmp(field, "real") == 0 || strcmp(field, "integer") == 0;
  int is_sym     = strcmp(sym, "symmetric") == 0;
  int is_gen     = strcmp(sym, "general") == 0;
  if (strcmp(fmt, "coordinate") != 0 || !(is_real || is_pattern) || !(is_sym || is_gen)) {
    fprintf(stderr, "oracle: unsupported Matrix Market type [%s %s %s]\n", fmt, field, sym);
    exit(EXIT_FAILURE);
  }
  int M = 0, N = 0, nz = 0;
  for (;;) {
    if (!fgets(line, sizeof line, f)) exit(EXIT_FAILURE);
    if (line[0] == '%') continue;
    if (sscanf(line, "%d %d %d", &M, &N, &nz) == 3) break;
  }
  mm_entry* e = (mm_entry*)xmalloc((size_t)nz * (is_sym ? 2 : 1) * sizeof *e);
  size_t cur  = 0;
  for (int i = 0; i < nz; i++) {
    int r, c;
    double v = 1.0;
    if (is_pattern) {
      if (fscanf(f, "%d %d", &r, &c) != 2) exit(EXIT_FAILURE);
    } else {
      if (fscanf(f, "%d %d %lg", &r, &c, &v) != 3) exit(EXIT_FAILURE);
    }
    r--;
    c--;
    e[cur].row = r, e[cur].col = c, e[cur].val = v, cur++;
    if (is_sym && r != c) e[cur].row = c, e[cur].col = r, e[cur].val = v, cur++;
  }
  fclose(f);
  mm_entry* tmp = (mm_entry*)xmalloc(cur * sizeof *tmp);
  mm_merge_sort(e, tmp, cur);
  free(tmp);
  *nrOut    = M;
  *countOut = cur;
  return e;
}

/* row split of file matrices: src/comm.c:35-38 (sizeOfRank), :347-361 */
static void rank_rows(int rank, int size, int N, int* start, int* stop)
{
  int cursor = 0;
  for (int i = 0; i <= rank; i++) {
    int n  = N / size + ((N % size > i) ? 1 : 0);
    *start = cursor;
    cursor += n;
    *stop = cursor - 1;
  }
}

orc_gmatrix* orc_mm_load_part(const char* path, int rank, int size)
{
  int M;
  size_t count;
  mm_entry* e = mm_read_all(path, &M, &count);
  int start = 0, stop = -1;
  rank_rows(rank, size, M, &start, &stop);
  orc_gmatrix* g = (orc_gmatrix*)xmalloc(sizeof *g);
  uint32_t nr    = (uint32_t)(stop - start + 1);
  g->rowPtr      = (uint32_t*)xmalloc(((size_t)nr + 1) * sizeof(uint32_t));
  memset(g->rowPtr, 0, ((size_t)nr + 1) * sizeof(uint32_t));
  size_t lo = 0;
  while (lo < count && e[lo].row < start) lo++;
  size_t hi = lo;
  while (hi < count && e[hi].row <= stop) hi++;
  size_t n = hi - lo;
  g->col   = (uint32_t*)xmalloc(n * sizeof(uint32_t));
  g->val   = (double*)xmalloc(n * sizeof(double));
  /* src/matrix.c:253-268: per-row counts -> prefix sums -> copy in order */
  for (size_t i = lo; i < hi; i++) g->rowPtr[e[i].row - start + 1]++;
  for (uint32_t r = 0; r < nr; r++) g->rowPtr[r + 1] += g->rowPtr[r];
  for (size_t i = lo; i < hi; i++) {
    g->col[i - lo] = (uint32_t)e[i].col;
    g->val[i - lo] = e[i].val;
  }
  free(e);
  g->nr = g->nc = nr;
  g->nnz = g->nnzTrue = (uint32_t)n;
  g->totalNr          = (uint32_t)M;
  g->totalNnz         = (uint32_t)count;
  g->startRow         = (uint32_t)start;
  g->stopRow          = (uint32_t)stop;
  g->generated        = 0;
  return g;
}

orc_gmatrix* orc_mm_load(const char* path) { return orc_mm_load_part(path, 0, 1); }

orc_gmatrix* orc_gm_from_arrays(uint32_t nr, uint32_t nc, const uint32_t* rowPtr,
                                const uint32_t* col, const double* val)
{
  orc_gmatrix* g = (orc_gmatrix*)xmalloc(sizeof *g);
  uint32_t nnz   = rowPtr[nr];
  g->rowPtr      = (uint32_t*)xmalloc(((size_t)nr + 1) * sizeof(uint32_t));
  g->col         = (uint32_t*)xmalloc((size_t)nnz * sizeof(uint32_t));
  g->val         = (double*)xmalloc((size_t)nnz * sizeof(double));
  memcpy(g->rowPtr, rowPtr, ((size_t)nr + 1) * sizeof(uint32_t));
  memcpy(g->col, col, (size_t)nnz * sizeof(uint32_t));
  memcpy(g->val, val, (size_t)nnz * sizeof(double));
  g->nr = nr, g->nc = nc, g->nnz = g->nnzTrue = nnz;
  g->totalNr = nr, g->totalNnz = nnz, g->startRow = 0, g->stopRow = nr ? nr - 1 : 0;
  g->generated = 0;
  return g;
}

void orc_gm_free(orc_gmatrix* g)
{
  if (!g) return;
  free(g->rowPtr);
  free(g->col);
  free(g->val);
  free(g);
}

/* ===================================================================== */
/* Partition + halo plan: src/comm.c:414-625 (commPartition),            */
/* :40-114 (buildIndexMapping), :116-182 (buildElementsToSend).          */
/* All P ranks are processed in one process, so the MPI exchanges of the  */
/* reference (Allgather of start rows :496, Dist_graph :540-579, the      */
/* Send/Irecv of wanted ids :134-161) become array lookups.               */
/* Externals: first-seen order (:452-473); owner = last rank whose start  */
/* row <= id (:505-520); local ids nr.. grouped by owner.  Groups are laid*/
/* out by ASCENDING owner (what MPI returns for sources[] and what rdispls*/
/* assume, :149-159); identical to the reference's first-seen-owner order */
/* on every input where the reference itself is consistent.               */
/* ===================================================================== */
typedef struct {
  uint32_t key;
  int value;
} hslot;

static size_t hash_u32(uint32_t k, size_t mask)
{
  uint64_t h = (uint64_t)k * 0x9E3779B97F4A7C15ull;
  return (size_t)(h >> 20) & mask;
}

orc_plan* orc_partition(orc_gmatrix** L, int P)
{
  orc_plan* plans = (orc_plan*)xmalloc((size_t)P * sizeof *plans);
  memset(plans, 0, (size_t)P * sizeof *plans);
  uint32_t* starts = (uint32_t*)xmalloc((size_t)P * sizeof(uint32_t));
  for (int r = 0; r < P; r++) starts[r] = L[r]->startRow;

  /* per-rank: external discovery + renumbering */
  int** extOwner = (int**)xmalloc((size_t)P * sizeof(int*));
  for (int r = 0; r < P; r++) {
    orc_gmatrix* A = L[r];
    orc_plan* c    = &plans[r];
    c->rank = r, c->size = P;
    size_t cap = 1 << 12;
    hslot* tab = (hslot*)xmalloc(cap * sizeof *tab);
    for (size_t i = 0; i < cap; i++) tab[i].value = -1;
    size_t extCap    = 1024;
    uint32_t* extIdx = (uint32_t*)xmalloc(extCap * sizeof(uint32_t));
    int extCount     = 0;
    /* step 1 (:452-473): first-seen list of external global ids */
    for (uint32_t i = 0; i < A->nr; i++)
      for (uint32_t j = A->rowPtr[i]; j < A->rowPtr[i + 1]; j++) {
        uint32_t cidx = A->col[j];
        if (cidx >= A->startRow && cidx <= A->stopRow) continue;
        size_t h = hash_u32(cidx, cap - 1);
        while (tab[h].value >= 0 && tab[h].key != cidx) h = (h + 1) & (cap - 1);
        if (tab[h].value >= 0) continue;
        tab[h].key = cidx, tab[h].value = extCount;
        if ((size_t)extCount == extCap) {
          extCap *= 2;
          extIdx = (uint32_t*)realloc(extIdx, extCap * sizeof(uint32_t));
        }
        extIdx[extCount++] = cidx;
        if ((size_t)extCount * 2 > cap) { /* grow + rehash */
          size_t ncap = cap * 4;
          hslot* nt   = (hslot*)xmalloc(ncap * sizeof *nt);
          for (size_t q = 0; q < ncap; q++) nt[q].value = -1;
          for (size_t q = 0; q < cap; q++)
            if (tab[q].value >= 0) {
              size_t hh = hash_u32(tab[q].key, ncap - 1);
              while (nt[hh].value >= 0) hh = (hh + 1) & (ncap - 1);
              nt[hh] = tab[q];
            }
          free(tab);
          tab = nt, cap = ncap;
        }
      }
    c->externalCount = extCount;
    /* step 2 (:505-533): owner of each external, per-owner counts */
    int* owner  = (int*)xmalloc((size_t)(extCount ? extCount : 1) * sizeof(int));
    int* perOwn = (int*)xmalloc((size_t)P * sizeof(int));
    memset(perOwn, 0, (size_t)P * sizeof(int));
    for (int i = 0; i < extCount; i++) {
      int o = 0;
      for (int j = P - 1; j >= 0; j--)
        if (starts[j] <= extIdx[i]) {
          o = j;
          break;
        }
      owner[i] = o;
      perOwn[o]++;
    }
    c->indegree = 0;
    for (int j = 0; j < P; j++) c->indegree += perOwn[j] > 0;
    c->sources    = (int*)xmalloc((size_t)(c->indegree + 1) * sizeof(int));
    c->recvCounts = (int*)xmalloc((size_t)(c->indegree + 1) * sizeof(int));
    c->rdispls    = (int*)xmalloc((size_t)(c->indegree + 1) * sizeof(int));
    int* ownBase  = (int*)xmalloc((size_t)P * sizeof(int));
    int cursor = 0, s = 0;
    for (int j = 0; j < P; j++) {
      ownBase[j] = cursor;
      if (perOwn[j] > 0) {
        c->sources[s] = j, c->recvCounts[s] = perOwn[j], c->rdispls[s] = cursor, s++;
        cursor += perOwn[j];
      }
    }
    /* step 3 (:60-110): local id = nr + (owner group base + first-seen rank in group) */
    int* localOf      = (int*)xmalloc((size_t)(extCount ? extCount : 1) * sizeof(int));
    c->externalGlobal = (uint32_t*)xmalloc((size_t)(extCount ? extCount : 1) * sizeof(uint32_t));
    int* fill         = (int*)xmalloc((size_t)P * sizeof(int));
    memset(fill, 0, (size_t)P * sizeof(int));
    for (int i = 0; i < extCount; i++) {
      int pos              = ownBase[owner[i]] + fill[owner[i]]++;
      localOf[i]           = (int)A->nr + pos;
      c->externalGlobal[pos] = extIdx[i];
    }
    for (uint32_t i = 0; i < A->nr; i++)
      for (uint32_t j = A->rowPtr[i]; j < A->rowPtr[i + 1]; j++) {
        uint32_t cidx = A->col[j];
        if (cidx >= A->startRow && cidx <= A->stopRow) {
          A->col[j] = cidx - A->startRow;
        } else {
          size_t h = hash_u32(cidx, cap - 1);
          while (tab[h].key != cidx || tab[h].value < 0) h = (h + 1) & (cap - 1);
          A->col[j] = (uint32_t)localOf[tab[h].value];
        }
      }
    A->nc = A->nr + (uint32_t)extCount; /* :616 */
    extOwner[r] = owner;
    free(tab), free(extIdx), free(perOwn), free(ownBase), free(localOf), free(fill);
  }

  /* step 4 (:116-182): what each rank must send = what the others asked of it,
   * in the asker's external order, destinations ascending. */
  for (int r = 0; r < P; r++) {
    orc_plan* c  = &plans[r];
    c->outdegree = 0;
    c->totalSendCount = 0;
    for (int d = 0; d < P; d++)
      for (int s = 0; s < plans[d].indegree; s++)
        if (plans[d].sources[s] == r) {
          c->outdegree++;
          c->totalSendCount += plans[d].recvCounts[s];
        }
    c->destinations   = (int*)xmalloc((size_t)(c->outdegree + 1) * sizeof(int));
    c->sendCounts     = (int*)xmalloc((size_t)(c->outdegree + 1) * sizeof(int));
    c->sdispls        = (int*)xmalloc((size_t)(c->outdegree + 1) * sizeof(int));
    c->elementsToSend = (int*)xmalloc((size_t)(c->totalSendCount + 1) * sizeof(int));
    int o = 0, cur = 0;
    for (int d = 0; d < P; d++)
      for (int s = 0; s < plans[d].indegree; s++)
        if (plans[d].sources[s] == r) {
          c->destinations[o] = d;
          c->sendCounts[o]   = plans[d].recvCounts[s];
          c->sdispls[o]      = cur;
          for (int i = 0; i < plans[d].recvCounts[s]; i++)
            c->elementsToSend[cur++] =
                (int)(plans[d].externalGlobal[plans[d].rdispls[s] + i] - L[r]->startRow);
          o++;
        }
  }
  for (int r = 0; r < P; r++) free(extOwner[r]);
  free(extOwner);
  free(starts);
  return plans;
}

void orc_plan_free(orc_plan* plans, int P)
{
  if (!plans) return;
  for (int r = 0; r < P; r++) {
    free(plans[r].sources), free(plans[r].recvCounts), free(plans[r].rdispls);
    free(plans[r].destinations), free(plans[r].sendCounts), free(plans[r].sdispls);
    free(plans[r].elementsToSend), free(plans[r].externalGlobal);
  }
  free(plans);
}

/* ===================================================================== */
/* Sell-C-sigma conversion: src/matrix-SCS.c:31-196, with the reference's */
/* defects fixed as DESIGN.md states: caller's C and sigma are honoured   */
/* (the reference clobbers them, :42-43) and nc keeps the externals (:38).*/
/* Layout is otherwise exactly the reference's: sigma-window descending   */
/* STABLE sort of row lengths over the padded row range (:61-79),         */
/* chunkLens = longest row of the chunk, chunkPtr = prefix of len*C       */
/* (:88-117), perms (:120-143), zero padding col 0 / val 0.0 (:146-155),  */
/* column-major fill inside the chunk keeping within-row order (:164-192).*/
/* ===================================================================== */
typedef struct {
  int index, count;
} rowlen;

static void rowlen_merge_sort(rowlen* a, rowlen* tmp, size_t n)
{
  if (n < 2) return;
  size_t h = n / 2;
  rowlen_merge_sort(a, tmp, h);
  rowlen_merge_sort(a + h, tmp, n - h);
  size_t i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].count > a[i].count) ? a[j++] : a[i++];
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, n * sizeof *a);
}

orc_scs* orc_convert_scs(const orc_gmatrix* g, uint32_t C, uint32_t sigma)
{
  orc_scs* m = (orc_scs*)xmalloc(sizeof *m);
  if (C == 0) C = 1;
  if (sigma == 0) sigma = 1;
  m->nr = g->nr, m->nc = g->nc, m->nnz = g->nnz, m->C = C, m->sigma = sigma;
  m->nChunks  = (g->nr + C - 1) / C;
  m->nrPadded = m->nChunks * C;
  rowlen* rl  = (rowlen*)xmalloc((size_t)(m->nrPadded + 1) * sizeof *rl);
  rowlen* tmp = (rowlen*)xmalloc((size_t)(sigma + 1) * sizeof *tmp);
  for (uint32_t i = 0; i < m->nrPadded; i++) {
    rl[i].index = (int)i;
    rl[i].count = i < g->nr ? (int)(g->rowPtr[i + 1] - g->rowPtr[i]) : 0;
  }
  for (uint32_t i = 0; i < m->nrPadded; i += sigma) {
    uint32_t stop = i + sigma < m->nrPadded ? i + sigma : m->nrPadded;
    rowlen_merge_sort(rl + i, tmp, stop - i);
  }
  free(tmp);
  m->chunkLens = (uint32_t*)xmalloc((size_t)(m->nChunks + 1) * sizeof(uint32_t));
  m->chunkPtr  = (uint32_t*)xmalloc((size_t)(m->nChunks + 1) * sizeof(uint32_t));
  uint32_t cp  = 0;
  for (uint32_t c = 0; c < m->nChunks; c++) {
    uint32_t mx = 0;
    for (uint32_t j = 0; j < C; j++)
      if ((uint32_t)rl[c * C + j].count > mx) mx = (uint32_t)rl[c * C + j].count;
    m->chunkLens[c] = mx;
    m->chunkPtr[c]  = cp;
    cp += mx * C;
  }
  m->nElems               = cp;
  m->chunkPtr[m->nChunks] = cp;
  m->oldToNewPerm = (uint32_t*)xmalloc((size_t)(g->nr + 1) * sizeof(uint32_t));
  m->newToOldPerm = (uint32_t*)xmalloc((size_t)(g->nr + 1) * sizeof(uint32_t));
  for (uint32_t i = 0; i < m->nrPadded; i++)
    if ((uint32_t)rl[i].index < g->nr) m->oldToNewPerm[rl[i].index] = i;
  /* NOTE: as in the reference (:132-143) newToOldPerm is indexed by the NEW
   * position, which can reach nrPadded-1 >= nr when padded rows sort ahead of
   * real ones; padded (empty) rows always sort last inside their window and the
   * last window is the only one that holds them, so new positions of real rows
   * stay < nr. */
  for (uint32_t i = 0; i < g->nr; i++) m->newToOldPerm[m->oldToNewPerm[i]] = i;
  free(rl);
  m->colInd = (uint32_t*)xmalloc((size_t)(m->nElems + 1) * sizeof(uint32_t));
  m->val    = (double*)xmalloc((size_t)(m->nElems + 1) * sizeof(double));
  for (uint32_t i = 0; i < m->nElems; i++) m->colInd[i] = 0, m->val[i] = 0.0;
  for (uint32_t i = 0; i < g->nr; i++) {
    uint32_t row  = m->oldToNewPerm[i];
    uint32_t base = m->chunkPtr[row / C] + row % C;
    uint32_t k    = 0;
    for (uint32_t j = g->rowPtr[i]; j < g->rowPtr[i + 1]; j++, k++) {
      m->colInd[base + k * C] = g->col[j];
      m->val[base + k * C]    = g->val[j];
    }
  }
  return m;
}

void orc_scs_free(orc_scs* s)
{
  if (!s) return;
  free(s->chunkPtr), free(s->chunkLens), free(s->colInd), free(s->val);
  free(s->oldToNewPerm), free(s->newToOldPerm);
  free(s);
}

/* ===================================================================== */
/* Kernels                                                               */
/* ===================================================================== */

/* src/matrix-CRS.c:46-65: sequential left-to-right sum per row */
void orc_spmv_crs(const orc_gmatrix* g, const double* x, double* y)
{
  for (uint32_t i = 0; i < g->nr; i++) {
    double sum = 0.0;
    for (uint32_t j = g->rowPtr[i]; j < g->rowPtr[i + 1]; j++) sum += g->val[j] * x[g->col[j]];
    y[i] = sum;
  }
}

/* src/matrix-SCS.c:198-228, literal: y[nrPadded] in permuted order */
void orc_spmv_scs_literal(const orc_scs* s, const double* x, double* y)
{
  uint32_t C  = s->C;
  double* tmp = (double*)xmalloc((size_t)C * sizeof(double));
  for (uint32_t c = 0; c < s->nChunks; c++) {
    for (uint32_t k = 0; k < C; k++) tmp[k] = 0.0;
    uint32_t off = s->chunkPtr[c];
    for (uint32_t j = 0; j < s->chunkLens[c]; j++)
      for (uint32_t k = 0; k < C; k++)
        tmp[k] += s->val[off + j * C + k] * x[s->colInd[off + j * C + k]];
    for (uint32_t k = 0; k < C; k++) y[c * C + k] = tmp[k];
  }
  free(tmp);
}

/* Same arithmetic, fixed output semantics (DESIGN.md "SCS semantics"):
 * row at permuted position q is stored to y[newToOldPerm[q]]; padded rows are
 * dropped, so y has nr entries and spMVM is y = A x for every sigma. */
void orc_spmv_scs(const orc_scs* s, const double* x, double* y)
{
  uint32_t C  = s->C;
  double* tmp = (double*)xmalloc((size_t)C * sizeof(double));
  for (uint32_t c = 0; c < s->nChunks; c++) {
    for (uint32_t k = 0; k < C; k++) tmp[k] = 0.0;
    uint32_t off = s->chunkPtr[c];
    for (uint32_t j = 0; j < s->chunkLens[c]; j++)
      for (uint32_t k = 0; k < C; k++)
        tmp[k] += s->val[off + j * C + k] * x[s->colInd[off + j * C + k]];
    for (uint32_t k = 0; k < C; k++) {
      uint32_t q = c * C + k;
      if (q < s->nr) y[s->newToOldPerm[q]] = tmp[k];
    }
  }
  free(tmp);
}

/* src/solver.c:16-39.  Under strict IEEE the three branches are bitwise equal
 * to the general form (1.0*x is exact); kept anyway to mirror the reference. */
void orc_waxpby(uint32_t n, double alpha, const double* x, double beta, const double* y,
                double* w)
{
  if (alpha == 1.0) {
    for (uint32_t i = 0; i < n; i++) w[i] = x[i] + beta * y[i];
  } else if (beta == 1.0) {
    for (uint32_t i = 0; i < n; i++) w[i] = alpha * x[i] + y[i];
  } else {
    for (uint32_t i = 0; i < n; i++) w[i] = alpha * x[i] + beta * y[i];
  }
}

/* src/solver.c:41-62, one thread: strictly sequential sum */
double orc_ddot_seq(uint32_t n, const double* x, const double* y)
{
  double sum = 0.0;
  for (uint32_t i = 0; i < n; i++) sum += x[i] * y[i];
  return sum;
}

/* "Exact" dot: Ogita-Rump-Oishi Dot2 -- every product split error-free (TwoProduct through fma), every
 * addition error-free (TwoSum), the error terms accumulated separately and added at the end: the result
 * is as good as a sum carried in twice the working precision and then rounded once.  For r.r (condition
 * number 1) and p.Ap (SPD: no cancellation to speak of) that IS the correctly rounded value up to a
 * relative 2^-53 * (1 + n * 2^-53 * cond): the yardstick both the reference's sequential sum and the GPU's
 * tree sum are measured against (tests/golden/cg_hist_exact.json, tests/test_gpu_cg.py).  One thread. */
double orc_ddot_exact(uint32_t n, const double* x, const double* y)
{
  double s = 0.0, c = 0.0;
  for (uint32_t i = 0; i < n; i++) {
    const double p  = x[i] * y[i];
    const double pe = fma(x[i], y[i], -p); /* x*y = p + pe exactly */
    const double t  = s + p;
    const double z  = t - s;
    const double se = (s - (t - z)) + (p - z); /* s + p = t + se exactly */
    s = t;
    c += se + pe;
  }
  return s + c;
}

/* Canonical reduction order of the HIP kernels (DESIGN.md "dot order").
 * Level 0: every aligned group of 64 consecutive elements is reduced by an xor
 * butterfly with offsets 1,2,4,8,16,32 (what 64 lanes do with shuffles).
 * Level 1: every aligned group of four such groups (256 elements) is added left to
 * right, ((g0+g1)+g2)+g3 -> one partial per 256 elements (what one workgroup of four
 * wavefronts does).  Missing tail elements / groups count as +0.0. */
static double group64(uint32_t g, uint32_t n, const double* x, const double* y)
{
  double v[64], t[64];
  for (uint32_t l = 0; l < 64; l++) {
    uint64_t i = (uint64_t)g * 64 + l;
    v[l]       = i < n ? x[i] * y[i] : 0.0;
  }
  for (uint32_t off = 1; off < 64; off <<= 1) {
    for (uint32_t l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
    memcpy(v, t, sizeof v);
  }
  return v[0];
}

void orc_ddot_partials(uint32_t n, const double* x, const double* y, double* partials)
{
  uint32_t m = (n + 255) / 256;
  for (uint32_t q = 0; q < m; q++)
    partials[q] = ((group64(4 * q, n, x, y) + group64(4 * q + 1, n, x, y)) + group64(4 * q + 2, n, x, y)) +
                  group64(4 * q + 3, n, x, y);
}

/* Level 2: 1024 (virtual) threads; thread t sums partials t, t+1024, ...
 * sequentially; each wave of 64 threads butterflies (1..32); the 16 wave sums
 * are added in wave order. */
double orc_reduce_final(uint32_t m, const double* q)
{
  double w[16];
  for (uint32_t wave = 0; wave < 16; wave++) {
    double v[64], t[64];
    for (uint32_t l = 0; l < 64; l++) {
      double s = 0.0;
      for (uint32_t i = wave * 64 + l; i < m; i += 1024) s += q[i];
      v[l] = s;
    }
    for (uint32_t off = 1; off < 64; off <<= 1) {
      for (uint32_t l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
      memcpy(v, t, sizeof v);
    }
    w[wave] = v[0];
  }
  double total = w[0];
  for (int i = 1; i < 16; i++) total += w[i];
  return total;
}

double orc_ddot_tree(uint32_t n, const double* x, const double* y)
{
  uint32_t m = (n + 255) / 256;
  double* q  = (double*)xmalloc((size_t)(m + 1) * sizeof(double));
  orc_ddot_partials(n, x, y, q);
  double r = orc_reduce_final(m, q);
  free(q);
  return r;
}

/* ===================================================================== */
/* CG: src/CGSolver.c:62-141 (solveCG), :19-38 (initVectors), :40-60      */
/* (solverCheckResidual); halo src/comm.c:627-651; all-reduce :653-662.   */
/* P ranks advance in lock step; the all-reduce adds the rank-local sums  */
/* either in rank order or as a pairwise tree.                            */
/* ===================================================================== */
static double reduce_ranks(double* v, int P, int rank_sum)
{
  if (rank_sum == 0 || P == 1) {
    double s = v[0];
    for (int r = 1; r < P; r++) s += v[r];
    return s;
  }
  /* recursive doubling on P ranks (P power of two) == pairwise tree */
  double t[64];
  int n = P;
  for (int r = 0; r < P; r++) t[r] = v[r];
  while (n > 1) {
    int h = 0;
    for (int r = 0; r + 1 < n; r += 2) t[h++] = t[r] + t[r + 1];
    if (n & 1) t[h++] = t[n - 1];
    n = h;
  }
  return t[0];
}

typedef struct {
  orc_gmatrix* g;
  orc_scs* s;
  double *r, *p, *Ap, *x, *b, *xexact;
} rank_state;

static void halo(rank_state* st, const orc_plan* plans, int P)
{
  if (P == 1 || !plans) return;
  /* pack on the sender (src/comm.c:635-638), deliver into x+nr of the
   * receiver at rdispls of that source (:640-648) */
  for (int d = 0; d < P; d++) {
    const orc_plan* pd = &plans[d];
    for (int s = 0; s < pd->indegree; s++) {
      int src              = pd->sources[s];
      const orc_plan* ps   = &plans[src];
      int o                = 0;
      while (ps->destinations[o] != d) o++;
      const int* el = ps->elementsToSend + ps->sdispls[o];
      double* dst   = st[d].p + st[d].g->nr + pd->rdispls[s];
      for (int i = 0; i < pd->recvCounts[s]; i++) dst[i] = st[src].p[el[i]];
    }
  }
}

static double tree_dot_in_storage_order(const rank_state* st, const double* a, const double* b)
{
  const uint32_t n = st->g->nr;
  int permuted     = 0;
  if (st->s)
    for (uint32_t i = 0; i < n && !permuted; i++) permuted = st->s->oldToNewPerm[i] != i;
  if (!permuted) return orc_ddot_tree(n, a, b);
  double* ta = (double*)xmalloc((size_t)n * sizeof(double));
  double* tb = (double*)xmalloc((size_t)n * sizeof(double));
  for (uint32_t q = 0; q < n; q++) ta[q] = a[st->s->newToOldPerm[q]], tb[q] = b[st->s->newToOldPerm[q]];
  double r = orc_ddot_tree(n, ta, tb);
  free(ta), free(tb);
  return r;
}

static void spmv_any(rank_state* st, int fmt)
{
  if (fmt == 0) orc_spmv_crs(st->g, st->p, st->Ap);
  else orc_spmv_scs(st->s, st->p, st->Ap);
}

int orc_cg(orc_gmatrix** L, const orc_plan* plans, int P, int fmt, uint32_t C, uint32_t sigma,
           int itermax, double eps, int dot_mode, int rank_sum, double* rr, int* n_rr,
           double* pAp, int* n_pAp, double** x_out, double* max_err)
{
  rank_state* st = (rank_state*)xmalloc((size_t)P * sizeof *st);
  double* loc    = (double*)xmalloc((size_t)P * sizeof(double));
  for (int q = 0; q < P; q++) {
    orc_gmatrix* g = L[q];
    st[q].g        = g;
    st[q].s        = fmt == 1 ? orc_convert_scs(g, C, sigma) : NULL;
    st[q].r        = (double*)xmalloc((size_t)g->nr * sizeof(double));
    st[q].p        = (double*)xmalloc((size_t)g->nc * sizeof(double));
    st[q].Ap       = (double*)xmalloc((size_t)g->nr * sizeof(double));
    st[q].x        = (double*)xmalloc((size_t)g->nr * sizeof(double));
    st[q].b        = (double*)xmalloc((size_t)g->nr * sizeof(double));
    st[q].xexact   = g->generated ? (double*)xmalloc((size_t)g->nr * sizeof(double)) : NULL;
    for (uint32_t i = 0; i < g->nc; i++) st[q].p[i] = 0.0;
    /* initVectors, src/CGSolver.c:25-36 */
    for (uint32_t i = 0; i < g->nr; i++) {
      int nnzrow = (int)(g->rowPtr[i + 1] - g->rowPtr[i]);
      st[q].x[i] = 0.0;
      if (st[q].xexact) {
        st[q].b[i]      = 27.0 - ((double)(nnzrow - 1));
        st[q].xexact[i] = 1.0;
      } else {
        st[q].b[i] = 1.0;
      }
    }
  }
  /* The HIP CG keeps its vectors in the SCS matrix's permuted row order (sigma > 1),
   * so its fixed-order dot runs over that order: restate it by gathering first. */
#define DOT(A, B, OUT)                                                              \
  do {                                                                              \
    for (int q_ = 0; q_ < P; q_++)                                                  \
      loc[q_] = dot_mode == 2 ? orc_ddot_exact(st[q_].g->nr, st[q_].A, st[q_].B)    \
                : dot_mode    ? tree_dot_in_storage_order(&st[q_], st[q_].A, st[q_].B) \
                              : orc_ddot_seq(st[q_].g->nr, st[q_].A, st[q_].B);     \
    (OUT) = reduce_ranks(loc, P, rank_sum);                                         \
  } while (0)

  int nrr = 0, npap = 0;
  double rtrans = 0.0, oldrtrans = 0.0, normr;
  /* prologue, :94-100 */
  for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].x, 0.0, st[q].x, st[q].p);
  halo(st, plans, P);
  for (int q = 0; q < P; q++) spmv_any(&st[q], fmt);
  for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].b, -1.0, st[q].Ap, st[q].r);
  DOT(r, r, rtrans);
  rr[nrr++] = rtrans;
  normr     = sqrt(rtrans);
  int k;
  for (k = 1; k < itermax && normr > eps; k++) { /* :107 */
    if (k == 1) {
      for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].r, 0.0, st[q].r, st[q].p);
    } else {
      oldrtrans = rtrans;
      DOT(r, r, rtrans);
      rr[nrr++]   = rtrans;
      double beta = rtrans / oldrtrans;
      for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].r, beta, st[q].p, st[q].p);
    }
    normr = sqrt(rtrans);
    halo(st, plans, P);
    for (int q = 0; q < P; q++) spmv_any(&st[q], fmt);
    double alpha = 0.0;
    DOT(p, Ap, alpha);
    pAp[npap++] = alpha;
    alpha       = rtrans / alpha;
    for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].x, alpha, st[q].p, st[q].x);
    for (int q = 0; q < P; q++) orc_waxpby(st[q].g->nr, 1.0, st[q].r, -alpha, st[q].Ap, st[q].r);
  }
#undef DOT
  *n_rr  = nrr;
  *n_pAp = npap;
  /* solverCheckResidual, :40-60 */
  double res = 0.0;
  for (int q = 0; q < P; q++)
    if (st[q].xexact)
      for (uint32_t i = 0; i < st[q].g->nr; i++) {
        double d = fabs(st[q].x[i] - st[q].xexact[i]);
        if (d > res) res = d;
      }
  if (max_err) *max_err = res;
  for (int q = 0; q < P; q++) {
    if (x_out) x_out[q] = st[q].x;
    else free(st[q].x);
    free(st[q].r), free(st[q].p), free(st[q].Ap), free(st[q].b), free(st[q].xexact);
    orc_scs_free(st[q].s);
  }
  free(st);
  free(loc);
  return k;
}

/* ===================================================================== */
/* cpu_baseline "port" leg: the same loops with the reference's OpenMP    */
/* pragmas (static schedule, src/solver.c:24-54, src/matrix-CRS.c:54).    */
/* Timing only -- parity never uses these.                                */
/* ===================================================================== */
static void spmv_crs_omp(const orc_gmatrix* g, const double* x, double* y)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)g->nr; i++) {
    double sum = 0.0;
    for (uint32_t j = g->rowPtr[i]; j < g->rowPtr[i + 1]; j++) sum += g->val[j] * x[g->col[j]];
    y[i] = sum;
  }
}

static void waxpby_omp(long n, double a, const double* x, double b, const double* y, double* w)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) w[i] = a * x[i] + b * y[i];
}

static double ddot_omp(long n, const double* x, const double* y)
{
  double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
  for (long i = 0; i < n; i++) sum += x[i] * y[i];
  return sum;
}

static int omp_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

double orc_time_spmv(orc_gmatrix* g, int reps, int* threads_used)
{
  double* x = (double*)xmalloc((size_t)g->nc * sizeof(double));
  double* y = (double*)xmalloc((size_t)g->nr * sizeof(double));
  for (uint32_t i = 0; i < g->nc; i++) x[i] = 1.0;
  spmv_crs_omp(g, x, y);
  double t0 = now_s();
  for (int k = 0; k < reps; k++) spmv_crs_omp(g, x, y);
  double t = now_s() - t0;
  free(x), free(y);
  if (threads_used) *threads_used = omp_threads();
  return t;
}

/* `iters` CG loop bodies (k>=2 shape: 2 ddot, 3 waxpby, 1 SpMV), 1 rank */
double orc_time_cg_iters(orc_gmatrix* g, int iters, int* threads_used)
{
  long n     = g->nr;
  double* r  = (double*)xmalloc((size_t)n * sizeof(double));
  double* p  = (double*)xmalloc((size_t)g->nc * sizeof(double));
  double* Ap = (double*)xmalloc((size_t)n * sizeof(double));
  double* x  = (double*)xmalloc((size_t)n * sizeof(double));
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) {
    x[i] = 0.0;
    r[i] = 27.0 - (double)((int)(g->rowPtr[i + 1] - g->rowPtr[i]) - 1);
    p[i] = r[i];
  }
  double rtrans = ddot_omp(n, r, r);
  spmv_crs_omp(g, p, Ap);
  double t0 = now_s();
  for (int k = 0; k < iters; k++) {
    double old  = rtrans;
    rtrans      = ddot_omp(n, r, r);
    double beta = rtrans / old;
    waxpby_omp(n, 1.0, r, beta, p, p);
    spmv_crs_omp(g, p, Ap);
    double alpha = rtrans / ddot_omp(n, p, Ap);
    waxpby_omp(n, 1.0, x, alpha, p, x);
    waxpby_omp(n, 1.0, r, -alpha, Ap, r);
  }
  double t = now_s() - t0;
  free(r), free(p), free(Ap), free(x);
  if (threads_used) *threads_used = omp_threads();
  return t;
}
