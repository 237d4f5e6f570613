/* sbh_flat.c -- a pointer-and-size surface over the C host side for callers that
 * cannot share C structs (the ctypes binding used by tests/ and bench.py).  It only
 * sequences the same calls the C driver makes: generate | read -> commPartition ->
 * convertMatrix; nothing is computed here.
 */
#include <stdlib.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

void sbh_comm_attach_halo(Comm* c, CG_UINT nr, const CG_UINT* oldToNewPerm);
void sbh_layout_crs(CRSMatrix* m, GMatrix* im);
void sbh_layout_scs(SCSMatrix* m, GMatrix* im);

typedef struct {
  int fmt; /* 0 CRS, 1 SCS */
  int generated;
  Parameter par;
  Comm comm;
  GMatrix gm;
  CRSMatrix crs;
  SCSMatrix scs;
  CG_UINT nnzTrue;
  double setup_s;
} sbh_problem;

sbh_problem* sbh_problem_create(const char* filename, int nx, int ny, int nz, int fmt, int C, int sigma,
    int rank, int size, int upload)
{
  /* upload = 0 builds the host layout only (needs no GPU): used by the CPU tests */
  sbh_problem* p = (sbh_problem*)calloc(1, sizeof *p);
  double t0      = getTimeStamp();
  p->fmt         = fmt;
  initParameter(&p->par);
  p->par.filename = strdup(filename);
  p->par.nx = nx, p->par.ny = ny, p->par.nz = nz;
  p->comm.rank = rank, p->comm.size = size;
  p->generated = strcmp(filename, "generate") == 0 || strcmp(filename, "generate7P") == 0;
  sbh_init_matrix(&p->comm, &p->par, &p->gm);
  p->nnzTrue = p->gm.rowPtr[p->gm.nr];
  commPartition(&p->comm, &p->gm);
  if (fmt == 0) {
    if (upload) sbh_convert_crs(&p->crs, &p->gm);
    else sbh_layout_crs(&p->crs, &p->gm);
  } else {
    p->scs.C = (CG_UINT)C, p->scs.sigma = (CG_UINT)sigma;
    if (upload) sbh_convert_scs(&p->scs, &p->gm);
    else sbh_layout_scs(&p->scs, &p->gm);
  }
  if (upload) sbh_comm_attach_halo(&p->comm, p->gm.nr, fmt == 1 ? p->scs.oldToNewPerm : NULL);
  p->setup_s = getTimeStamp() - t0;
  return p;
}

void* sbh_problem_matrix(sbh_problem* p) { return p->fmt == 0 ? p->crs.dev : p->scs.dev; }
void* sbh_problem_halo(sbh_problem* p) { return p->comm.dev; }
double sbh_problem_setup_seconds(sbh_problem* p) { return p->setup_s; }

/* which: 0 nr, 1 nc, 2 nnz (reference's), 3 true nnz, 4 totalNr, 5 totalNnz, 6 startRow,
 * 7 stopRow, 8 C, 9 sigma, 10 nChunks, 11 nrPadded, 12 nElems, 13 externalCount,
 * 14 totalSendCount, 15 indegree, 16 outdegree */
unsigned sbh_problem_scalar(sbh_problem* p, int which)
{
  switch (which) {
  case 0: return p->gm.nr;
  case 1: return p->gm.nc;
  case 2: return p->gm.nnz;
  case 3: return p->nnzTrue;
  case 4: return p->gm.totalNr;
  case 5: return p->gm.totalNnz;
  case 6: return p->gm.startRow;
  case 7: return p->gm.stopRow;
  case 8: return p->scs.C;
  case 9: return p->scs.sigma;
  case 10: return p->scs.nChunks;
  case 11: return p->scs.nrPadded;
  case 12: return p->scs.nElems;
  case 13: return (unsigned)p->comm.externalCount;
  case 14: return (unsigned)p->comm.totalSendCount;
  case 15: return (unsigned)p->comm.indegree;
  case 16: return (unsigned)p->comm.outdegree;
  default: return 0;
  }
}

/* which: 0 rowPtr, 1 rowNnz, 2 crs.colInd, 3 scs.chunkPtr, 4 scs.chunkLens, 5 scs.colInd,
 * 6 scs.oldToNewPerm, 7 scs.newToOldPerm, 8 elementsToSend, 9 sources, 10 recvCounts,
 * 11 rdispls, 12 destinations, 13 sendCounts, 14 sdispls, 15 externalGlobal */
const void* sbh_problem_array(sbh_problem* p, int which)
{
  switch (which) {
  case 0: return p->gm.rowPtr;
  case 1: return p->fmt == 0 ? p->crs.rowNnz : p->scs.rowNnz;
  case 2: return p->crs.colInd;
  case 3: return p->scs.chunkPtr;
  case 4: return p->scs.chunkLens;
  case 5: return p->scs.colInd;
  case 6: return p->scs.oldToNewPerm;
  case 7: return p->scs.newToOldPerm;
  case 8: return p->comm.elementsToSend;
  case 9: return p->comm.sources;
  case 10: return p->comm.recvCounts;
  case 11: return p->comm.rdispls;
  case 12: return p->comm.destinations;
  case 13: return p->comm.sendCounts;
  case 14: return p->comm.sdispls;
  case 15: return p->comm.externalGlobal;
  default: return NULL;
  }
}

/* `file.mtx` -> `file.bmx` next to it, as the driver's -c option (src/main.c:41-52); no device needed */
void sbh_convert_mtx_to_bmx(const char* mtxFilename)
{
  Comm c;
  memset(&c, 0, sizeof c);
  c.size = 1;
  char* name = strdup(mtxFilename);
  sbh_write_bin_matrix(&c, name);
  free(name);
}

/* one rank's row slice of a .bmx file, before commPartition (global column ids) */
typedef struct {
  GMatrix gm;
} sbh_gm;

sbh_gm* sbh_bmx_read(const char* filename, int rank, int size)
{
  sbh_gm* g = (sbh_gm*)calloc(1, sizeof *g);
  Comm c;
  memset(&c, 0, sizeof c);
  c.rank = rank, c.size = size;
  char* name = strdup(filename);
  matrixBinRead(&g->gm, &c, name);
  free(name);
  return g;
}

unsigned sbh_gm_scalar(sbh_gm* g, int which)
{
  switch (which) {
  case 0: return g->gm.nr;
  case 1: return g->gm.nnz;
  case 2: return g->gm.startRow;
  case 3: return g->gm.stopRow;
  case 4: return g->gm.totalNr;
  default: return g->gm.totalNnz;
  }
}

void sbh_gm_copy(sbh_gm* g, unsigned* rowPtr, unsigned* col, double* val)
{
  for (CG_UINT i = 0; i <= g->gm.nr; i++) rowPtr[i] = g->gm.rowPtr[i];
  for (CG_UINT i = 0; i < g->gm.nnz; i++) col[i] = g->gm.entries[i].col, val[i] = g->gm.entries[i].val;
}

void sbh_gm_free(sbh_gm* g)
{
  if (!g) return;
  free(g->gm.rowPtr), free(g->gm.entries), free(g);
}

const double* sbh_problem_values(sbh_problem* p) { return p->fmt == 0 ? p->crs.val : p->scs.val; }

void sbh_problem_gm_entries(sbh_problem* p, unsigned* col, double* val)
{
  for (CG_UINT i = 0; i < p->nnzTrue; i++) col[i] = p->gm.entries[i].col, val[i] = p->gm.entries[i].val;
}

/* b and xexact of initVectors (src/CGSolver.c:25-36); xexact may be NULL */
int sbh_problem_rhs(sbh_problem* p, double* b, double* xexact)
{
  const CG_UINT* len = p->fmt == 0 ? p->crs.rowNnz : p->scs.rowNnz;
  for (CG_UINT i = 0; i < p->gm.nr; i++) {
    b[i] = p->generated ? 27.0 - ((double)((int)len[i] - 1)) : 1.0;
    if (xexact) xexact[i] = 1.0;
  }
  return p->generated;
}

void sbh_problem_free(sbh_problem* p)
{
  if (!p) return;
  if (p->comm.dev) sb_halo_free((sb_halo*)p->comm.dev);
  if (p->crs.dev || p->scs.dev) sb_matrix_free((sb_matrix*)(p->fmt == 0 ? p->crs.dev : p->scs.dev));
  free(p->gm.rowPtr), free(p->gm.entries);
  if (p->fmt == 0) free(p->crs.rowPtr), free(p->crs.colInd), free(p->crs.val), free(p->crs.rowNnz);
  else
    free(p->scs.chunkPtr), free(p->scs.chunkLens), free(p->scs.colInd), free(p->scs.val),
        free(p->scs.oldToNewPerm), free(p->scs.newToOldPerm), free(p->scs.rowNnz);
  free(p->comm.elementsToSend), free(p->comm.sources), free(p->comm.recvCounts), free(p->comm.rdispls);
  free(p->comm.destinations), free(p->comm.sendCounts), free(p->comm.sdispls), free(p->comm.externalGlobal);
  free(p);
}
