/* dropin.c -- the link-time slot.  Compiled twice, -DCRS and -DSCS, into
 * libsparsebench_crs.so / libsparsebench_scs.so: each exports convertMatrix, spMVM and
 * solveCG for ITS Matrix typedef, exactly as the reference links one matrix-<FMT>.o
 * (reference Makefile:20,32-34; src/matrix.h:14-22,57; src/solver.h:11-13).
 */
#include <stdlib.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

int sbh_solve_cg_perm(Comm* comm, Parameter* param, void* dev_matrix, CG_UINT nr, const CG_UINT* rowNnz,
    const CG_UINT* oldToNewPerm);

#if defined(CRS)

void convertMatrix(Matrix* m, GMatrix* im) { sbh_convert_crs(m, im); }

int solveCG(Comm* comm, Parameter* param, Matrix* m)
{
  return sbh_solve_cg_perm(comm, param, m->dev, m->nr, m->rowNnz, NULL);
}

#elif defined(SCS)

/* The reference's driver never sets C / sigma (src/main.c:173-174) and its
 * convertMatrix overwrites them (src/matrix-SCS.c:42-43).  Here: SPARSEBENCH_C /
 * SPARSEBENCH_SIGMA win if set; otherwise plausible caller values are honoured;
 * otherwise C = 64 (one wavefront per chunk), sigma = 1. */
static CG_UINT pick(const char* env, CG_UINT given, CG_UINT lo, CG_UINT hi, CG_UINT dflt)
{
  const char* v = getenv(env);
  if (v && atoi(v) > 0) return (CG_UINT)atoi(v);
  if (given >= lo && given <= hi) return given;
  return dflt;
}

void convertMatrix(Matrix* m, GMatrix* im)
{
  m->C     = pick("SPARSEBENCH_C", m->C, 1, 4096, 64);
  m->sigma = pick("SPARSEBENCH_SIGMA", m->sigma, 1, 1u << 24, 1);
  sbh_convert_scs(m, im);
}

int solveCG(Comm* comm, Parameter* param, Matrix* m)
{
  return sbh_solve_cg_perm(comm, param, m->dev, m->nr, m->rowNnz, m->oldToNewPerm);
}

#else
#error "compile with -DCRS or -DSCS"
#endif

void sbh_print_banner(Comm* c, const char* fmt);
void commPrintBanner(Comm* c) { sbh_print_banner(c, FMT); } /* src/comm.c:185-250: names the build's format */

void spMVM(Matrix* m, const CG_FLOAT* restrict x, CG_FLOAT* restrict y)
{
  sbh_spmv(m->dev, m->nr, m->nc, x, y);
}

/* src/comm.h:55 (VERBOSE-build diagnostic): scalars of the converted matrix */
void commMatrixDump(Comm* c, Matrix* m)
{
  FILE* f = c->logFile ? c->logFile : stdout;
  fprintf(f, "Matrix (%s): rank %d, nr %u nc %u nnz %u totalNr %u totalNnz %u rows %u..%u\n", FMT, c->rank, m->nr,
      m->nc, m->nnz, m->totalNr, m->totalNnz, m->startRow, m->stopRow);
#if defined(SCS)
  fprintf(f, "C %u sigma %u nrPadded %u nChunks %u nElems %u\n", m->C, m->sigma, m->nrPadded, m->nChunks, m->nElems);
#endif
  fflush(f);
}
