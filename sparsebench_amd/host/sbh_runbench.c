/* sbh_runbench.c -- kernel micro-benchmarks: ddot, waxpby and spMVM in isolation.
 *
 * The reference ships benchmarks/runBenchmarks.c as an empty stub with the TODO
 * "single core bench ddot, waxpby, spMVM" (benchmarks/runBenchmarks.c:1-5); this is that
 * program for the HIP path, driving the kernels through the reference-shaped API with
 * vectors resident in HBM.  Byte conventions are the reference profiler's
 * (src/profiler.c:19-22, src/main.c:187-189) next to the true-nnz figures of DESIGN.md.
 *
 * usage: runBenchmarks-<FMT>-HIP [-x nx -y ny -z nz] [-m file.mtx] [-i reps] [-C c -s sigma]
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <unistd.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

static double time_region(void (*fn)(void*), void* ctx, int reps)
{
  fn(ctx); /* warm-up */
  void* a = sb_event_create();
  void* b = sb_event_create();
  sb_event_record(a);
  for (int i = 0; i < reps; i++) fn(ctx);
  sb_event_record(b);
  double s = 1e-3 * sb_event_elapsed_ms(a, b) / reps;
  sb_event_destroy(a), sb_event_destroy(b);
  return s;
}

typedef struct {
  Matrix* m;
  CG_UINT n;
  double *x, *y, *w, *scalar;
} ctx_t;

static void run_spmv(void* p) { ctx_t* c = (ctx_t*)p; spMVM(c->m, c->x, c->y); }
static void run_waxpby(void* p) { ctx_t* c = (ctx_t*)p; waxpby(c->n, 1.0, c->y, 0.5, c->w, c->w); }
static void run_ddot(void* p) { ctx_t* c = (ctx_t*)p; sb_ddot_async(c->n, c->y, c->w, c->scalar); }

int main(int argc, char** argv)
{
  Comm comm;
  Parameter param;
  commInit(&comm, argc, argv);
  initParameter(&param);
  param.nx = param.ny = param.nz = 128;
  int reps = 100, opt;
  unsigned scsC = 64, scsSigma = 1;
  while ((opt = getopt(argc, argv, "m:x:y:z:i:C:s:")) != -1) switch (opt) {
    case 'm': param.filename = optarg; break;
    case 'x': param.nx = atoi(optarg); break;
    case 'y': param.ny = atoi(optarg); break;
    case 'z': param.nz = atoi(optarg); break;
    case 'i': reps = atoi(optarg); break;
    case 'C': scsC = (unsigned)atoi(optarg); break;
    case 's': scsSigma = (unsigned)atoi(optarg); break;
    default: fprintf(stderr, "unknown option\n"); return 1;
    }
  commPrintBanner(&comm);
  GMatrix g;
  sbh_init_matrix(&comm, &param, &g);
  const double nnzTrue = (double)g.rowPtr[g.nr];
  commPartition(&comm, &g);
  Matrix m;
  memset(&m, 0, sizeof m);
#ifdef SCS
  m.C = scsC, m.sigma = scsSigma;
#else
  (void)scsC, (void)scsSigma;
#endif
  convertMatrix(&m, &g);

  ctx_t c;
  c.m = &m, c.n = g.nr;
  c.x      = (double*)sb_malloc((size_t)g.nc * sizeof(double));
  c.y      = (double*)sb_malloc((size_t)g.nr * sizeof(double));
  c.w      = (double*)sb_malloc((size_t)g.nr * sizeof(double));
  c.scalar = (double*)sb_malloc(sizeof(double));
  double* ones = (double*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)g.nc + 1) * sizeof(double));
  for (CG_UINT i = 0; i < g.nc; i++) ones[i] = 1.0;
  sb_h2d(c.x, ones, (size_t)g.nc * sizeof(double));
  sb_h2d(c.w, ones, (size_t)g.nr * sizeof(double));

  const double tS = time_region(run_spmv, &c, reps);
  const double tW = time_region(run_waxpby, &c, reps);
  const double tD = time_region(run_ddot, &c, reps);
  const double n  = (double)g.nr;
  const double bytesSpmvRef = 12.0 * (double)g.nnz; /* the reference's convention */
  const double bytesSpmv    = sb_matrix_spmv_bytes((const sb_matrix*)m.dev);
  const double bytesStream  = sb_matrix_stream_bytes((const sb_matrix*)m.dev);
  if (commIsMaster(&comm)) {
    printf("rows %u  stored nonzeros %.0f  repetitions %d\n", g.nr, nnzTrue, reps);
    printf(HLINE);
    printf("kernel     time(us)   GB/s(algorithmic)   GB/s(reference convention)   GFlop/s\n");
    printf("spMVM   %10.2f %12.1f %20.1f %22.1f\n", 1e6 * tS, 1e-9 * bytesSpmv / tS, 1e-9 * bytesSpmvRef / tS,
        1e-9 * 2.0 * nnzTrue / tS);
    printf("waxpby  %10.2f %12.1f %20.1f %22.1f\n", 1e6 * tW, 1e-9 * 24.0 * n / tW, 1e-9 * 24.0 * n / tW,
        1e-9 * 3.0 * n / tW);
    printf("ddot    %10.2f %12.1f %20.1f %22.1f\n", 1e6 * tD, 1e-9 * 16.0 * n / tD, 1e-9 * 16.0 * n / tD,
        1e-9 * 2.0 * n / tD);
    printf(HLINE);
    printf("spMVM moves %.1f MB per launch (reference layout: %.1f MB; pack level %d)\n", 1e-6 * bytesStream,
        1e-6 * bytesSpmv, sb_matrix_pack_level((const sb_matrix*)m.dev));
  }
  commFinalize(&comm);
  return EXIT_SUCCESS;
}
