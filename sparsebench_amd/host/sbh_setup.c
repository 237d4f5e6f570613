/* sbh_setup.c -- host-side problem setup: HPCG stencil generator and Matrix Market
 * reader.  Same inputs/outputs as the reference (matrixGenerate src/matrix.c:30-121,
 * MMMatrixRead :123-229, matrixConvertfromMM :231-269), written from scratch:
 * the generator is a two-pass, OpenMP-parallel closed-form fill (the reference's
 * single-threaded triple loop took 7 s at 128^3), the reader does its
 * own header parsing and orders entries with two stable counting sorts.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdlib.h>

#include "sparsebench/sparsebench.h"

/* ------------------------------------------------------------------------------
 * HPCG generator.  Rank r owns the brick of global z-planes [r*nz, (r+1)*nz) of an
 * nx x ny x (nz*size) grid.  Row = grid point; its neighbours in the 3x3x3 box that
 * exist in the global grid give the entries, ordered (dz, dy, dx) ascending --
 * which is ascending global column id.  Diagonal 27.0, others -1.0.  Columns are
 * GLOBAL ids here; commPartition renumbers them.
 * m->nnz / totalNnz keep the reference's upper bound 27*rows (src/matrix.c:35-38,
 * :117-120) because its profiler rates are defined on it; the true count is
 * rowPtr[nr].
 * ------------------------------------------------------------------------------ */
static inline int span_lo(int i) { return i > 0 ? -1 : 0; }
static inline int span_hi(int i, int n) { return i < n - 1 ? 1 : 0; }

void matrixGenerate(GMatrix* m, Parameter* p, int rank, int size, bool use_7pt_stencil)
{
  const int nx = p->nx, ny = p->ny, nz = p->nz;
  const long plane   = (long)nx * ny;
  const long localNr = plane * nz;
  const long totalNr = localNr * size;
  const long gzTotal = (long)nz * size;
  const long start   = localNr * rank;

  if (rank == 0) {
    printf("Generate %s matrix with ", use_7pt_stencil ? "7pt" : "27pt");
    printf("%.2e total rows and %.2e nonzeros\n", (double)totalNr, (double)(27 * localNr));
  }

  m->rowPtr = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, (size_t)(localNr + 1) * sizeof(CG_UINT));

  /* pass 1: row lengths in closed form */
#pragma omp parallel for schedule(static)
  for (long iz = 0; iz < nz; iz++) {
    const long gz = (long)rank * nz + iz;
    const int cz  = 1 + (gz > 0) + (gz < gzTotal - 1);
    for (int iy = 0; iy < ny; iy++) {
      const int cy = 1 + (iy > 0) + (iy < ny - 1);
      for (int ix = 0; ix < nx; ix++) {
        const int cx = 1 + (ix > 0) + (ix < nx - 1);
        const int n  = use_7pt_stencil ? 1 + (cz - 1) + (cy - 1) + (cx - 1) : cz * cy * cx;
        m->rowPtr[iz * plane + (long)iy * nx + ix + 1] = (CG_UINT)n;
      }
    }
  }
  m->rowPtr[0] = 0;
  for (long i = 0; i < localNr; i++) m->rowPtr[i + 1] += m->rowPtr[i];
  const size_t nnzTrue = m->rowPtr[localNr];
  m->entries           = (Entry*)sbh_alloc_host(ARRAY_ALIGNMENT, (nnzTrue + 1) * sizeof(Entry));

  /* pass 2: fill */
#pragma omp parallel for schedule(static)
  for (long iz = 0; iz < nz; iz++) {
    const long gz = (long)rank * nz + iz;
    for (int iy = 0; iy < ny; iy++)
      for (int ix = 0; ix < nx; ix++) {
        const long row = iz * plane + (long)iy * nx + ix;
        const long me  = start + row;
        Entry* e       = m->entries + m->rowPtr[row];
        for (int dz = (gz > 0 ? -1 : 0); dz <= (gz < gzTotal - 1 ? 1 : 0); dz++)
          for (int dy = span_lo(iy); dy <= span_hi(iy, ny); dy++)
            for (int dx = span_lo(ix); dx <= span_hi(ix, nx); dx++) {
              if (use_7pt_stencil && dz * dz + dy * dy + dx * dx > 1) continue;
              const long col = me + dz * plane + (long)dy * nx + dx;
              e->col         = (CG_UINT)col;
              e->val         = (col == me) ? 27.0 : -1.0;
              e++;
            }
      }
  }

  m->startRow = (CG_UINT)start;
  m->stopRow  = (CG_UINT)(start + localNr - 1);
  m->totalNr  = (CG_UINT)totalNr;
  m->totalNnz = (CG_UINT)(27 * totalNr);
  m->nr       = (CG_UINT)localNr;
  m->nc       = (CG_UINT)localNr;
  m->nnz      = (CG_UINT)(27 * localNr);
}

/* ------------------------------------------------------------------------------
 * Matrix Market coordinate files.  Accepted: real | integer | pattern, general |
 * symmetric (what src/matrix.c:139-170 lets through).  Indices become 0-based,
 * symmetric files get their off-diagonal mirror, and the entries end up ordered by
 * (row, col) with file order kept among duplicates -- the effect of the reference's
 * sort by column followed by a stable sort by row (:219-228).
 * ------------------------------------------------------------------------------ */
static void lower(char* s)
{
  for (; *s; s++) *s = (char)tolower((unsigned char)*s);
}

static void counting_sort(const MMEntry* in, MMEntry* out, size_t n, int keys, int by_row)
{
  size_t* start = (size_t*)calloc((size_t)keys + 1, sizeof(size_t));
  for (size_t i = 0; i < n; i++) start[(by_row ? in[i].row : in[i].col) + 1]++;
  for (int k = 0; k < keys; k++) start[k + 1] += start[k];
  for (size_t i = 0; i < n; i++) out[start[by_row ? in[i].row : in[i].col]++] = in[i];
  free(start);
}

void MMMatrixRead(MMMatrix* m, char* filename)
{
  FILE* f = fopen(filename, "r");
  if (f == NULL) {
    printf("Unable to open file.\n");
    exit(EXIT_FAILURE);
  }
  char line[4200], tag[64], object[64], format[64], field[64], symmetry[64];
  if (!fgets(line, sizeof line, f) ||
      sscanf(line, "%63s %63s %63s %63s %63s", tag, object, format, field, symmetry) != 5 ||
      strcmp(tag, "%%MatrixMarket") != 0) {
    printf("Could not process Matrix Market banner.\n");
    exit(EXIT_FAILURE);
  }
  lower(object), lower(format), lower(field), lower(symmetry);
  const bool pattern   = strcmp(field, "pattern") == 0;
  const bool numeric   = strcmp(field, "real") == 0 || strcmp(field, "integer") == 0;
  const bool symmetric = strcmp(symmetry, "symmetric") == 0;
  const bool general   = strcmp(symmetry, "general") == 0;
  if (strcmp(object, "matrix") != 0 || strcmp(format, "coordinate") != 0 || !(numeric || pattern)) {
    fprintf(stderr, "Sorry, this application does not support ");
    fprintf(stderr, "Market Market type: [%s %s %s %s]\n", object, format, field, symmetry);
    exit(EXIT_FAILURE);
  }
  if (!(symmetric || general)) {
    printf("The matrix market file provided is not supported.\n Reason :\n");
    printf(" * matrix has to be symmetric\n");
    exit(EXIT_FAILURE);
  }
  int M = 0, N = 0, nz = 0;
  for (;;) {
    if (!fgets(line, sizeof line, f)) exit(EXIT_FAILURE);
    if (line[0] == '%') continue;
    if (sscanf(line, "%d %d %d", &M, &N, &nz) == 3) break;
  }
  printf("Read matrix %s with %d non zeroes and %d rows\n", filename, nz, M);

  const size_t cap = (size_t)nz * (symmetric ? 2 : 1);
  MMEntry* a       = (MMEntry*)sbh_alloc_host(ARRAY_ALIGNMENT, (cap + 1) * sizeof(MMEntry));
  size_t n         = 0;
  for (int i = 0; i < nz; i++) {
    int r, c;
    double v = 1.0;
    int got  = pattern ? fscanf(f, "%d %d", &r, &c) : fscanf(f, "%d %d %lg", &r, &c, &v);
    if (got != (pattern ? 2 : 3)) {
      fprintf(stderr, "Matrix Market file %s: entry %d is malformed\n", filename, i + 1);
      exit(EXIT_FAILURE);
    }
    r--, c--;
    if (r < 0 || r >= M || c < 0 || c >= N) {
      fprintf(stderr, "Matrix Market file %s: entry %d out of range\n", filename, i + 1);
      exit(EXIT_FAILURE);
    }
    a[n].row = r, a[n].col = c, a[n].val = v, n++;
    if (symmetric && r != c) a[n].row = c, a[n].col = r, a[n].val = v, n++;
  }
  fclose(f);

  MMEntry* b = (MMEntry*)sbh_alloc_host(ARRAY_ALIGNMENT, (n + 1) * sizeof(MMEntry));
  counting_sort(a, b, n, N > M ? N : M, 0);
  counting_sort(b, a, n, M, 1);
  free(b);

  m->entries = a;
  m->nr      = M;
  m->nnz     = (int)n;
  m->count   = n;
}

/* src/matrix.c:231-269: rows of [startRow, stopRow] -> row pointer + (col,val) */
void matrixConvertfromMM(MMMatrix* mm, GMatrix* m)
{
  m->startRow = (CG_UINT)mm->startRow;
  m->stopRow  = (CG_UINT)mm->stopRow;
  m->totalNr  = (CG_UINT)mm->totalNr;
  m->totalNnz = (CG_UINT)mm->totalNnz;
  m->nr       = (CG_UINT)mm->nr;
  m->nc       = (CG_UINT)mm->nr;
  m->nnz      = (CG_UINT)mm->nnz;
  m->entries  = (Entry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nnz + 1) * sizeof(Entry));
  m->rowPtr   = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nr + 1) * sizeof(CG_UINT));
  memset(m->rowPtr, 0, ((size_t)m->nr + 1) * sizeof(CG_UINT));
  for (size_t i = 0; i < mm->count; i++) m->rowPtr[mm->entries[i].row - mm->startRow + 1]++;
  for (CG_UINT r = 0; r < m->nr; r++) m->rowPtr[r + 1] += m->rowPtr[r];
  for (size_t i = 0; i < mm->count; i++) { /* entries are already (row, col) ordered */
    m->entries[i].col = (CG_UINT)mm->entries[i].col;
    m->entries[i].val = (CG_FLOAT)mm->entries[i].val;
  }
}
