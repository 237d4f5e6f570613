/* sbh_main.c -- the benchmark driver: SparseBench's command line and output, HIP
 * hot path.  Built per format like the reference executable (sparseBench-<FMT>-HIP).
 * Follows src/main.c:83-230 call for call; differences:
 *   - every rank reads a .mtx itself (no MPI scatter), see commDistributeMatrix;
 *   - SCS gets -C <chunk height> and -s <sigma> (the reference has no flag, and no
 *     working SCS path);
 *   - -t spmv is the reference's own loop (allocate() + host loops + PROFILE(spMVM)); the allocation hook hands out
 *     HBM-resident, host-visible vectors, so the loop times the kernel (host/sbh_base.c);
 *   - .bmx files (-c <file.mtx> writes one, -m <file.bmx> loads one) go through plain
 *     POSIX I/O instead of MPI-IO (sbh_binfile.c); SB_BMX_FP64=1 keeps fp64 values.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdlib.h>
#include <unistd.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

typedef enum { CG = 0, SPMV, GMRES, CHEBFD } bench_type;

static const char* kHelp =
    "Usage: sparseBench [options]\n\n"
    "Options:\n"
    "  -h         Show this help text\n"
    "  -f <parameter file>   Load options from a parameter file\n"
    "  -c <file name>   Convert MM matrix to binary matrix file (.bmx).\n"
    "  -m <matrix>   Load a matrix market (.mtx) or binary (.bmx) file\n"
    "  -t <bench type>   Benchmark type, can be cg or spmv. Default cg.\n"
    "  -x <int>   Size in x for generated matrix, ignored if MM file is loaded. Default 100.\n"
    "  -y <int>   Size in y for generated matrix, ignored if MM file is loaded. Default 100.\n"
    "  -z <int>   Size in z for generated matrix, ignored if MM file is loaded. Default 100.\n"
    "  -i <int>   Number of solver iterations. Default 150.\n"
    "  -e <float>  Convergence criteria epsilon. Default 0.0.\n"
    "  -C <int>   Sell-C-sigma chunk height (SCS build). Default 64.\n"
    "  -s <int>   Sell-C-sigma sorting scope (SCS build). Default 1.\n";

int main(int argc, char** argv)
{
  Parameter param;
  Comm comm;
  commInit(&comm, argc, argv);
  initParameter(&param);
  int type = CG, opt;
  unsigned scsC = 64, scsSigma = 1;
  opterr = 0;
  while ((opt = getopt(argc, argv, "hc:t:f:m:x:y:z:i:e:C:s:")) != -1) switch (opt) {
    case 'h':
      if (commIsMaster(&comm)) printf("%s", kHelp);
      commAbort(&comm, "");
      break;
    case 'c':
      sbh_write_bin_matrix(&comm, optarg); /* src/main.c:107-110 */
      commAbort(&comm, "Finish write matrix");
      break;
    case 'f': readParameter(&param, optarg); break;
    case 'm': param.filename = optarg; break;
    case 't':
      if (strcmp(optarg, "cg") == 0) type = CG;
      else if (strcmp(optarg, "spmv") == 0) type = SPMV;
      else {
        printf("Unknown solver type %s\n", optarg);
        return 1;
      }
      break;
    case 'x': param.nx = atoi(optarg); break;
    case 'y': param.ny = atoi(optarg); break;
    case 'z': param.nz = atoi(optarg); break;
    case 'i': param.itermax = atoi(optarg); break;
    case 'e': param.eps = atof(optarg); break;
    case 'C': scsC = (unsigned)atoi(optarg); break;
    case 's': scsSigma = (unsigned)atoi(optarg); break;
    default:
      if (isprint(optopt)) fprintf(stderr, "Unknown option `-%c'.\n", optopt);
      else fprintf(stderr, "Unknown option character `\\x%x'.\n", optopt);
      return 1;
    }
  for (int i = optind; i < argc; i++) printf("Non-option argument %s\n", argv[i]);

  commPrintBanner(&comm);

  double ts;
  GMatrix m;
  double timeStart = getTimeStamp();
  sbh_init_matrix(&comm, &param, &m);
  commPartition(&comm, &m);
  Matrix sm;
  memset(&sm, 0, sizeof sm);
#ifdef SCS
  sm.C = scsC, sm.sigma = scsSigma;
#else
  (void)scsC, (void)scsSigma;
#endif
  convertMatrix(&sm, &m);
  commBarrier();
  double timeStop = getTimeStamp();
  if (commIsMaster(&comm)) printf("Setup took %.2fs\n", timeStop - timeStart);

  size_t factorFlops[NUMREGIONS] = { 0 }, factorWords[NUMREGIONS] = { 0 };
  factorFlops[DDOT] = factorFlops[WAXPBY] = m.totalNr; /* src/main.c:181-190 */
  factorWords[DDOT] = factorWords[WAXPBY] = sizeof(CG_FLOAT) * (size_t)m.totalNr;
  factorFlops[SPMVM] = m.totalNnz;
  factorWords[SPMVM] = (sizeof(CG_FLOAT) + sizeof(CG_UINT)) * (size_t)m.totalNnz;
  profilerInit(factorFlops, factorWords);

  int k = 0;
  if (type == CG) {
    if (commIsMaster(&comm)) printf("Test type: CG\n");
    k = solveCG(&comm, &param, &sm);
  } else {
    if (commIsMaster(&comm)) printf("Test type: SPMVM\n");
    /* exactly the reference's loop (src/main.c:205-215): vectors from the allocation hook, filled by host loops, spMVM under
     * PROFILE.  allocate() hands out HBM-resident, host-visible memory (host/sbh_base.c), so the loop times the kernel. */
    CG_FLOAT* x = (CG_FLOAT*)allocate(ARRAY_ALIGNMENT, (size_t)m.nc * sizeof(CG_FLOAT));
    CG_FLOAT* y = (CG_FLOAT*)allocate(ARRAY_ALIGNMENT, (size_t)m.nr * sizeof(CG_FLOAT));
    for (CG_UINT i = 0; i < m.nc; i++) x[i] = (CG_FLOAT)1.0; /* (the reference leaves the halo tail uninitialised) */
    for (CG_UINT i = 0; i < m.nr; i++) y[i] = (CG_FLOAT)1.0;
    for (k = 1; k < param.itermax; k++) {
      PROFILE(SPMVM, spMVM(&sm, x, y));
    }
  }
  profilerPrint(&comm, k);
  profilerFinalize();
  commFinalize(&comm);
  return EXIT_SUCCESS;
}
