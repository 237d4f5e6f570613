/* ref_mpi_log.c -- ddot logger for the full MPI reference build (oracle/_ref/sb_ref_mpi).
 *
 * TEST INFRASTRUCTURE ONLY; ours, no reference code.  Linked with
 * -Wl,--wrap=ddot: rank 0 appends "<rr|pAp> %.17e" per ddot call to $DDOT_LOG,
 * so tests/golden/make_golden.py can capture multi-rank residual histories at
 * full precision (the reference prints %E only, src/CGSolver.c:118-120).
 */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

void __real_ddot(const unsigned n, const double* restrict x, const double* restrict y,
                 double* restrict result);

void __wrap_ddot(const unsigned n, const double* restrict x, const double* restrict y,
                 double* restrict result)
{
  __real_ddot(n, x, y, result);
  int rank = 0;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  const char* path = getenv("DDOT_LOG");
  if (rank == 0 && path) {
    FILE* f = fopen(path, "a");
    if (f) {
      fprintf(f, "%s %.17e\n", x == y ? "rr" : "pAp", *result);
      fclose(f);
    }
  }
}
