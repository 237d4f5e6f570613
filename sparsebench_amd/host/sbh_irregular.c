/* sbh_irregular.c -- a deterministic, irregular, SPD finite-element-like matrix: the committed
 * STAND-IN for BASELINE.json configs[4] (SuiteSparse Janna/Flan_1565 is not available offline:
 * SURVEY.md section 7 "Hard parts").  It enters the driver exactly where a Matrix Market file
 * would (src/main.c:54-81 initMatrix -> GMatrix with global column ids, rows split over ranks by
 * the file rule src/comm.c:35-38), under the file name "irregular":
 *
 *     sparseBench-CRS-HIP -m irregular -x 80 -y 80 -z 80        (1 536 000 rows, ~95 M nonzeros)
 *
 * What Flan_1565 stresses and this reproduces (3-D mesh, 3 unknowns per node, 1.56 M rows, 117 M
 * nonzeros, rows of 24..81 entries, structurally symmetric, SPD):
 *   - 3x3-block rows: node v couples to node u with a dense 3x3 block;
 *   - row lengths varying > 4x: nodes fall into three density classes (70 % / 20 % / 10 %), an
 *     edge of the 27-point node neighbourhood survives with the smaller of its two nodes' weights;
 *   - a few % far couplings (~5 % of the nonzeros): every node starts 0..2 edges to nodes anywhere in
 *     the mesh (the transposed entry is generated as well), which no banded window contains;
 *   - values: ~2e6 distinct values of either sign (no value dictionary applies), symmetric (A = A^T),
 *     strictly diagonally dominant with a positive diagonal => SPD, so CG converges on it (condition
 *     number ~40; with b = 1 the residual stays a normal number for > 600 iterations).
 * No random-number state: every decision is a fixed 64-bit mix of (seed, node, node), so any rank
 * can generate any row, and the union of the ranks' slices IS the one-rank matrix.  All values are
 * integers scaled by 2^-21 and sums stay below 2^31 * 2^-21, hence exact in fp64 whatever the order.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>

#include "sparsebench/sparsebench.h"

#define IRR_SEED 0x5BA15E0FF1A9ull

static inline uint64_t mix64(uint64_t z)
{ /* splitmix64 finaliser */
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t mix3(uint64_t a, uint64_t b, uint64_t c) { return mix64(mix64(mix64(IRR_SEED ^ a) ^ b) ^ c); }

/* density class of a node -> the weight an edge needs to survive, in 1/1024 */
static inline uint32_t node_weight(uint64_t v)
{
  const uint32_t c = (uint32_t)(mix3(1, v, 0) % 100u);
  return c < 70u ? 1024u : c < 90u ? 614u : 256u; /* 1.0 / 0.6 / 0.25 */
}
static inline int edge_kept(uint64_t a, uint64_t b)
{ /* a < b; symmetric by construction */
  const uint32_t wa = node_weight(a), wb = node_weight(b);
  return (uint32_t)(mix3(2, a, b) & 1023u) < (wa < wb ? wa : wb);
}
/* far edges node w starts: 0, 1 or 2 (56 % / 38 % / 6 %): every edge has two ends, so a node sees
 * one far neighbour on average among its ~20 -- "a few %" of the couplings */
static inline int far_count(uint64_t w)
{
  const uint32_t r = (uint32_t)(mix3(3, w, 0) & 15u);
  return r < 9u ? 0 : r < 15u ? 1 : 2;
}
static inline uint64_t far_target(uint64_t w, int k, uint64_t N) { return mix3(4, w, (uint64_t)k) % N; }

/* value of entry (i, j) of the 3x3 block coupling nodes a < b, as seen from a; from b it is (j, i) */
static inline double block_val(uint64_t a, uint64_t b, int i, int j)
{
  const uint64_t h = mix3(5 + (uint64_t)(3 * i + j), a, b);
  const double mag = (double)(1u + (uint32_t)(h >> 44)) * 0x1p-21; /* (1 .. 2^20) * 2^-21: (0, 0.5] */
  return (h & 7u) == 0u ? mag : -mag; /* 1 in 8 positive: A * 1 is not a constant vector (b = 1 is no eigenvector) */
}

typedef struct {
  int nx, ny, nz;
  uint64_t N;
  /* far edges by TARGET node: inPtr[N+1], inSrc[] (sources of the far edges that end in a node) */
  uint32_t* inPtr;
  uint32_t* inSrc;
} irr_mesh;

static int grid_adjacent(const irr_mesh* g, uint64_t a, uint64_t b)
{ /* b in the 3x3x3 box around a (including a itself) */
  const long plane = (long)g->nx * g->ny;
  const long az = (long)(a / plane), ay = (long)((a % plane) / g->nx), ax = (long)(a % g->nx);
  const long bz = (long)(b / plane), by = (long)((b % plane) / g->nx), bx = (long)(b % g->nx);
  return labs(az - bz) <= 1 && labs(ay - by) <= 1 && labs(ax - bx) <= 1;
}
/* the far edge (w -> its k-th target) exists unless it duplicates a mesh edge or is a loop */
static inline int far_valid(const irr_mesh* g, uint64_t w, uint64_t t) { return t != w && !grid_adjacent(g, w, t); }

static int cmp_u32(const void* a, const void* b)
{
  const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : x > y;
}

/* neighbours of node v (itself included), ascending and unique; returns how many (<= 27 + far) */
static int node_neighbours(const irr_mesh* g, uint64_t v, uint32_t* out, int cap)
{
  const long plane = (long)g->nx * g->ny;
  const int vz = (int)(v / plane), vy = (int)((v % plane) / g->nx), vx = (int)(v % g->nx);
  int n = 0;
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int z = vz + dz, y = vy + dy, x = vx + dx;
        if (z < 0 || z >= g->nz || y < 0 || y >= g->ny || x < 0 || x >= g->nx) continue;
        const uint64_t u = (uint64_t)z * plane + (uint64_t)y * g->nx + x;
        if (u == v || edge_kept(u < v ? u : v, u < v ? v : u)) out[n++] = (uint32_t)u;
      }
  const int nMesh = n;
  for (int k = 0; k < far_count(v); k++) {
    const uint64_t t = far_target(v, k, g->N);
    if (far_valid(g, v, t) && n < cap) out[n++] = (uint32_t)t;
  }
  for (uint32_t i = g->inPtr[v]; i < g->inPtr[v + 1] && n < cap; i++) out[n++] = g->inSrc[i];
  if (n > nMesh) { /* far ends: sort everything, drop duplicates (u -> v and v -> u, or two edges to one node) */
    qsort(out, (size_t)n, sizeof(uint32_t), cmp_u32);
    int m = 0;
    for (int i = 0; i < n; i++)
      if (m == 0 || out[m - 1] != out[i]) out[m++] = out[i];
    n = m;
  }
  if (n >= cap) {
    fprintf(stderr, "irregular: node %llu has more than %d neighbours\n", (unsigned long long)v, cap - 1);
    exit(EXIT_FAILURE);
  }
  return n;
}

static void mesh_build(irr_mesh* g, int nx, int ny, int nz)
{
  g->nx = nx, g->ny = ny, g->nz = nz;
  g->N  = (uint64_t)nx * ny * nz;
  g->inPtr = (uint32_t*)calloc(g->N + 2, sizeof(uint32_t));
  for (uint64_t w = 0; w < g->N; w++)
    for (int k = 0; k < far_count(w); k++) {
      const uint64_t t = far_target(w, k, g->N);
      if (far_valid(g, w, t)) g->inPtr[t + 1]++;
    }
  for (uint64_t v = 0; v < g->N; v++) g->inPtr[v + 1] += g->inPtr[v];
  g->inSrc       = (uint32_t*)malloc(((size_t)g->inPtr[g->N] + 1) * sizeof(uint32_t));
  uint32_t* fill = (uint32_t*)malloc((g->N + 1) * sizeof(uint32_t));
  memcpy(fill, g->inPtr, (g->N + 1) * sizeof(uint32_t));
  for (uint64_t w = 0; w < g->N; w++) /* ascending w: every target's source list is ascending */
    for (int k = 0; k < far_count(w); k++) {
      const uint64_t t = far_target(w, k, g->N);
      if (far_valid(g, w, t)) g->inSrc[fill[t]++] = (uint32_t)w;
    }
  free(fill);
}

#define IRR_MAXNB 96 /* 27 mesh neighbours + far ends (a node is the target of a handful at most) */

/* Fills m with this rank's rows of the 3*nx*ny*nz-row matrix (global column ids, ascending within a
 * row), split like a file's rows (src/comm.c:35-38).  nnz/totalNnz are true counts (a file's are). */
void sbh_matrix_generate_irregular(GMatrix* m, Parameter* p, int rank, int size)
{
  irr_mesh g;
  mesh_build(&g, p->nx, p->ny, p->nz);
  const uint64_t totalNr = 3ull * g.N;
  if (totalNr >= 0xFFFFFFFFull) {
    fprintf(stderr, "irregular: %llu rows do not fit CG_UINT\n", (unsigned long long)totalNr);
    exit(EXIT_FAILURE);
  }
  const uint64_t base = totalNr / (uint64_t)size, extra = totalNr % (uint64_t)size;
  const uint64_t first = (uint64_t)rank * base + ((uint64_t)rank < extra ? (uint64_t)rank : extra);
  const uint64_t nr    = base + ((uint64_t)rank < extra ? 1u : 0u);
  m->rowPtr = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, (size_t)(nr + 1) * sizeof(CG_UINT));

  /* pass 1: row lengths (3 per neighbour node) */
#pragma omp parallel for schedule(dynamic, 4096)
  for (long r = 0; r < (long)nr; r++) {
    uint32_t nb[IRR_MAXNB];
    m->rowPtr[r + 1] = 3u * (CG_UINT)node_neighbours(&g, (first + (uint64_t)r) / 3u, nb, IRR_MAXNB);
  }
  m->rowPtr[0] = 0;
  uint64_t total = 0;
  for (uint64_t r = 0; r < nr; r++) {
    total += m->rowPtr[r + 1];
    if (total > 0xFFFFFFFFull) {
      fprintf(stderr, "irregular: more than 2^32 nonzeros on one rank\n");
      exit(EXIT_FAILURE);
    }
    m->rowPtr[r + 1] = (CG_UINT)total;
  }
  m->entries = (Entry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)total + 1) * sizeof(Entry));

  /* pass 2: entries.  Row = unknown i of node v; block (v, u) contributes columns 3u .. 3u+2. */
#pragma omp parallel for schedule(dynamic, 4096)
  for (long r = 0; r < (long)nr; r++) {
    const uint64_t row = first + (uint64_t)r, v = row / 3u;
    const int i        = (int)(row % 3u);
    uint32_t nb[IRR_MAXNB];
    const int n = node_neighbours(&g, v, nb, IRR_MAXNB);
    Entry* e    = m->entries + m->rowPtr[r];
    Entry* diag = NULL;
    double off  = 0.0; /* sum of |off-diagonal| of this row: exact (multiples of 2^-21, < 2^10) */
    for (int q = 0; q < n; q++) {
      const uint64_t u = nb[q];
      for (int j = 0; j < 3; j++, e++) {
        e->col = (CG_UINT)(3u * u + (uint64_t)j);
        if (u == v && j == i) {
          diag = e;
          continue;
        }
        if (u == v) e->val = block_val(v, v, i < j ? i : j, i < j ? j : i); /* symmetric diagonal block */
        else if (v < u) e->val = block_val(v, u, i, j);
        else e->val = block_val(u, v, j, i); /* transpose of what u's row holds */
        off += e->val < 0.0 ? -e->val : e->val;
      }
    }
    /* strict diagonal dominance (margin 2^-4): positive definite.  Measured at 16^3 nodes: spectrum
     * [0.69, 25.7]; r.r falls by ~1e-16 per 50 CG iterations and stays a normal number for > 600 */
    diag->val = 0x1p-4 + off;
  }
  /* nc = local rows, as the reference's generator and matrixConvertfromMM leave it: columns stay global
   * until commPartition, which renumbers them and adds the externals (src/comm.c:616) */
  m->nr = (CG_UINT)nr, m->nc = (CG_UINT)nr;
  m->startRow = (CG_UINT)first, m->stopRow = (CG_UINT)(first + nr - 1);
  m->totalNr  = (CG_UINT)totalNr;
  m->nnz      = (CG_UINT)total;
  /* the global count, for the profiler's rate convention (src/main.c:187-189): every rank can count it */
  uint64_t all = 0;
  if (size == 1) all = total;
  else {
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : all)
    for (long v = 0; v < (long)g.N; v++) {
      uint32_t nb[IRR_MAXNB];
      all += 9ull * (uint64_t)node_neighbours(&g, (uint64_t)v, nb, IRR_MAXNB);
    }
  }
  free(g.inPtr), free(g.inSrc);
  m->totalNnz = (CG_UINT)(all > 0xFFFFFFFFull ? 0xFFFFFFFFull : all);
  if (rank == 0)
    printf("Generate irregular FE-like matrix (Flan_1565 stand-in) with %.2e total rows and %.2e nonzeros\n",
        (double)totalNr, (double)all);
}
