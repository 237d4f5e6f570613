/* ref_shim.c -- test harness linked INTO the reference build (oracle/_ref/).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.
 * It is compiled together with the reference's own sources (where they lie,
 * /root/reference/src) by oracle/build_ref.sh, and
 *   - intercepts ddot via the linker (-Wl,--wrap=ddot) so every dot product the
 *     reference's solveCG computes is recorded at full precision
 *     (the reference prints residuals only at %E, src/CGSolver.c:118-120);
 *   - exposes a flat, ctypes-friendly surface over the reference's struct API
 *     (src/matrix.h, src/solver.h) so Python never has to know struct layouts.
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "comm.h"
#include "matrix.h"
#include "parameter.h"
#include "solver.h"

#define SBREF_MAXHIST 65536
static double g_hist[SBREF_MAXHIST];
static int g_kind[SBREF_MAXHIST]; /* 0 = r.r (x==y), 1 = p.Ap */
static int g_nhist = 0;

void __real_ddot(const CG_UINT n, const CG_FLOAT* restrict x, const CG_FLOAT* restrict y,
                 CG_FLOAT* restrict result);

void __wrap_ddot(const CG_UINT n, const CG_FLOAT* restrict x, const CG_FLOAT* restrict y,
                 CG_FLOAT* restrict result)
{
  __real_ddot(n, x, y, result);
  if (g_nhist < SBREF_MAXHIST) {
    g_hist[g_nhist] = *result;
    g_kind[g_nhist] = (x == y) ? 0 : 1;
    g_nhist++;
  }
}

int sbref_hist_len(void) { return g_nhist; }
double sbref_hist_val(int i) { return g_hist[i]; }
int sbref_hist_kind(int i) { return g_kind[i]; }
void sbref_hist_reset(void) { g_nhist = 0; }

static GMatrix g_gm;
static Matrix g_m;
static Parameter g_par;
static Comm g_comm;

/* generate (filename "generate"/"generate7P") or read a .mtx, then convert */
void sbref_setup(const char* filename, int nx, int ny, int nz, int scsC, int scsSigma)
{
  memset(&g_gm, 0, sizeof g_gm);
  memset(&g_m, 0, sizeof g_m);
  g_comm.rank = 0, g_comm.size = 1, g_comm.logFile = NULL;
  g_par.filename = strdup(filename);
  g_par.nx = nx, g_par.ny = ny, g_par.nz = nz, g_par.itermax = 150, g_par.eps = 0.0;
  if (strcmp(filename, "generate") == 0) {
    matrixGenerate(&g_gm, &g_par, 0, 1, false);
  } else if (strcmp(filename, "generate7P") == 0) {
    matrixGenerate(&g_gm, &g_par, 0, 1, true);
  } else {
    MMMatrix mm, mml;
    memset(&mm, 0, sizeof mm), memset(&mml, 0, sizeof mml);
    MMMatrixRead(&mm, g_par.filename);
    commDistributeMatrix(&g_comm, &mm, &mml);
    mml.totalNr = mm.nr, mml.totalNnz = mm.nnz; /* left unset by the serial path */
    matrixConvertfromMM(&mml, &g_gm);
  }
  commPartition(&g_comm, &g_gm);
#ifdef SCS
  g_m.C = (CG_UINT)scsC, g_m.sigma = (CG_UINT)scsSigma;
#else
  (void)scsC, (void)scsSigma;
#endif
  convertMatrix(&g_m, &g_gm);
}

unsigned sbref_nr(void) { return g_gm.nr; }
unsigned sbref_nc(void) { return g_gm.nc; }
unsigned sbref_nnz(void) { return g_gm.nnz; }
unsigned sbref_nnz_true(void) { return g_gm.rowPtr[g_gm.nr]; }
unsigned sbref_total_nr(void) { return g_gm.totalNr; }
unsigned sbref_total_nnz(void) { return g_gm.totalNnz; }
const unsigned* sbref_rowptr(void) { return g_gm.rowPtr; }
void sbref_entries(unsigned* col, double* val)
{
  unsigned n = g_gm.rowPtr[g_gm.nr];
  for (unsigned i = 0; i < n; i++) col[i] = g_gm.entries[i].col, val[i] = g_gm.entries[i].val;
}

void sbref_spmv(const double* x, double* y) { spMVM(&g_m, x, y); }
void sbref_waxpby(unsigned n, double a, const double* x, double b, const double* y, double* w)
{
  waxpby(n, a, x, b, y, w);
}
double sbref_ddot(unsigned n, const double* x, const double* y)
{
  double r = 0.0;
  __real_ddot(n, x, y, &r);
  return r;
}
int sbref_solve_cg(int itermax, double eps)
{
  g_par.itermax = itermax, g_par.eps = eps;
  g_nhist = 0;
  return solveCG(&g_comm, &g_par, &g_m);
}

#ifdef SCS
unsigned sbref_scs_field(int which)
{
  switch (which) {
  case 0: return g_m.C;
  case 1: return g_m.sigma;
  case 2: return g_m.nChunks;
  case 3: return g_m.nrPadded;
  case 4: return g_m.nElems;
  default: return 0;
  }
}
const unsigned* sbref_scs_array(int which)
{
  switch (which) {
  case 0: return g_m.chunkPtr;
  case 1: return g_m.chunkLens;
  case 2: return g_m.colInd;
  case 3: return g_m.oldToNewPerm;
  case 4: return g_m.newToOldPerm;
  default: return NULL;
  }
}
const double* sbref_scs_val(void) { return g_m.val; }
#endif
