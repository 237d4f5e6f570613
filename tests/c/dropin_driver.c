/* dropin_driver.c -- a caller written ONLY against the reference-shaped API
 * (include/sparsebench/sparsebench.h): the sequence of src/main.c:164-225 plus direct
 * calls of spMVM / waxpby / ddot with HOST vectors, as the reference's -t spmv mode and
 * its (stale) unit tests do.  Prints values at full precision for the pytest side to
 * compare with the oracle.  Built twice: -DCRS and -DSCS.
 */
#include <stdlib.h>

#include "sparsebench/sparsebench.h"

int main(int argc, char** argv)
{
  Comm comm;
  Parameter param;
  commInit(&comm, argc, argv);
  initParameter(&param);
  param.nx = param.ny = param.nz = 12;
  param.itermax               = 25;
  if (argc > 1) param.filename = argv[1];
  GMatrix m;
  if (strcmp(param.filename, "generate") == 0) {
    matrixGenerate(&m, &param, comm.rank, comm.size, false);
  } else {
    MMMatrix mm, local;
    memset(&mm, 0, sizeof mm), memset(&local, 0, sizeof local);
    MMMatrixRead(&mm, param.filename);
    commDistributeMatrix(&comm, &mm, &local);
    matrixConvertfromMM(&local, &m);
  }
  commPartition(&comm, &m);
  Matrix sm;
  memset(&sm, 0, sizeof sm);
#ifdef SCS
  sm.C = 64, sm.sigma = 128;
#endif
  convertMatrix(&sm, &m);

  /* host vectors through the reference signatures */
  CG_FLOAT* x = (CG_FLOAT*)allocate(ARRAY_ALIGNMENT, m.nc * sizeof(CG_FLOAT));
  CG_FLOAT* y = (CG_FLOAT*)allocate(ARRAY_ALIGNMENT, m.nr * sizeof(CG_FLOAT));
  CG_FLOAT* w = (CG_FLOAT*)allocate(ARRAY_ALIGNMENT, m.nr * sizeof(CG_FLOAT));
  for (CG_UINT i = 0; i < m.nc; i++) x[i] = 1.0 + 0.001 * (double)(i % 97);
  spMVM(&sm, x, y);
  for (CG_UINT i = 0; i < m.nr; i++) printf("y %u %.17e\n", i, y[i]);
  waxpby(m.nr, 1.0, y, -0.5, x, w);
  for (CG_UINT i = 0; i < m.nr; i += 37) printf("w %u %.17e\n", i, w[i]);
  waxpby(m.nr, 2.0, w, 1.0, y, w); /* w aliases an input, as src/CGSolver.c:114 does */
  for (CG_UINT i = 0; i < m.nr; i += 37) printf("v %u %.17e\n", i, w[i]);
  CG_FLOAT d1 = 0.0, d2 = 0.0;
  ddot(m.nr, w, y, &d1);
  ddot(m.nr, y, y, &d2);
  printf("dot %.17e %.17e\n", d1, d2);

  size_t ff[NUMREGIONS] = { 0 }, fw[NUMREGIONS] = { 0 };
  ff[DDOT] = ff[WAXPBY] = m.totalNr, fw[DDOT] = fw[WAXPBY] = sizeof(CG_FLOAT) * (size_t)m.totalNr;
  ff[SPMVM] = m.totalNnz, fw[SPMVM] = 12 * (size_t)m.totalNnz;
  profilerInit(ff, fw);
  int k = solveCG(&comm, &param, &sm);
  printf("k %d\n", k);
  profilerPrint(&comm, k);
  profilerFinalize();
  commFinalize(&comm);
  return EXIT_SUCCESS;
}
