/* sparsebench.h -- the reference-shaped C API of the hot path, backed by the HIP layer.
 *
 * SparseBench picks its matrix format at compile time (-DCRS | -DSCS) and links
 * exactly one matrix-<FMT>.o (reference Makefile:20,32-34; src/matrix.h:14-22):
 * that link-time slot is its plugin boundary.  This header declares the same
 * symbols with the same signatures, so a driver written against the reference's
 * solver.h / matrix.h / comm.h compiles against it unchanged and links
 * libsparsebench_<fmt>.so instead of the reference objects.
 *
 * Differences a caller can observe, all additive:
 *   - Matrix gains two trailing members (`dev`, `rowNnz`) filled by convertMatrix;
 *   - Comm always carries the halo-plan members (the reference hides them behind
 *     _MPI) plus a trailing device handle;
 *   - SCS: the caller's C and sigma are honoured (the reference overwrites them
 *     with 1, src/matrix-SCS.c:42-43) and nc keeps the external columns (:38).
 *
 * Citations are paths in the reference tree.
 */
#ifndef SPARSEBENCH_H
#define SPARSEBENCH_H
#include <stdbool.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types: src/util.h:35-53 (defaults of config.mk:7-8) --------------- */
#ifndef CG_FLOAT
#define CG_FLOAT double
#endif
#ifndef CG_UINT
#define CG_UINT unsigned int
#endif
#define PRECISION_STRING "double"
#define UINT_STRING      "unsigned int"
#ifndef ARRAY_ALIGNMENT
#define ARRAY_ALIGNMENT 64 /* config.mk:11 */
#endif
#define HLINE "----------------------------------------------------------------------\n"
/* src/util.h:13-33 */
#ifndef MIN
#define MIN(x, y) ((x) < (y) ? (x) : (y))
#endif
#ifndef MAX
#define MAX(x, y) ((x) > (y) ? (x) : (y))
#endif
#ifndef ABS
#define ABS(a) ((a) >= 0 ? (a) : -(a))
#endif
#ifndef IS_EQUAL
#define IS_EQUAL(a, b) (strcmp((a), (b)) == 0)
#endif
#ifndef MAXLINE
#define MAXLINE 4096
#endif
/* src/util.h:55, src/util.c:11-32: "<name up to its last dot><newEnding>", malloc'ed */
char* changeFileEnding(char* filename, char* newEnding);
#define MAX_EXTERNAL 6000000 /* src/comm.h:16 -- kept for source compatibility only */

/* ---- run-time parameters: src/parameter.h:9-18 ---------------------------------- */
typedef struct {
  char* filename;
  int nx, ny, nz;
  int itermax;
  double eps;
} Parameter;

void initParameter(Parameter*);
void readParameter(Parameter*, const char*);
void printParameter(Parameter*);

/* ---- general matrix + Matrix Market staging: src/matrix.h:24-49 ----------------- */
typedef struct {
  CG_UINT col;
  CG_FLOAT val;
} Entry;

typedef struct {
  CG_UINT nr, nc, nnz;
  CG_UINT totalNr, totalNnz;
  CG_UINT startRow, stopRow;
  CG_UINT* rowPtr;
  Entry* entries;
} GMatrix;

typedef struct {
  int row;
  int col;
  double val;
} MMEntry;

typedef struct {
  size_t count;
  int nr, nnz;
  int totalNr, totalNnz;
  int startRow, stopRow;
  MMEntry* entries;
} MMMatrix;

/* ---- format-specific matrices ----------------------------------------------------- */
/* src/CRSMatrix.h:9-16 + trailing device members */
typedef struct {
  CG_UINT nr, nc, nnz;
  CG_UINT totalNr, totalNnz;
  CG_UINT startRow, stopRow;
  CG_UINT* rowPtr;
  CG_UINT* colInd;
  CG_FLOAT* val;
  void* dev;       /* sb_matrix* in HBM (include/sbhip.h) */
  CG_UINT* rowNnz; /* nonzeros per row, for initVectors (src/CGSolver.c:27) */
} CRSMatrix;

/* src/SCSMatrix.h:13-27 + trailing device members */
typedef struct {
  CG_UINT nr, nc, nnz;
  CG_UINT totalNr, totalNnz;
  CG_UINT startRow, stopRow;
  CG_UINT* colInd;
  CG_FLOAT* val;
  CG_UINT C, sigma;
  CG_UINT nrPadded, nChunks;
  CG_UINT nElems;
  CG_UINT* chunkPtr;
  CG_UINT* chunkLens;
  CG_UINT* oldToNewPerm;
  CG_UINT* newToOldPerm;
  void* dev;
  CG_UINT* rowNnz;
} SCSMatrix;

typedef struct { /* src/SCSMatrix.h:29-32 */
  int index;
  int count;
} SellCSigmaPair;

#if defined(CRS)
typedef CRSMatrix Matrix;
#define FMT "CRS"
#elif defined(SCS)
typedef SCSMatrix Matrix;
#define FMT "SCS"
#endif

/* ---- communication: src/comm.h:25-46 ------------------------------------------------ */
enum op { MAX = 0, SUM };

typedef struct {
  int rank;
  int size;
  FILE* logFile;
  int externalCount;
  int totalSendCount;
  int* elementsToSend;
  int indegree;
  int outdegree;
  int* sources;
  int* recvCounts;
  int* rdispls;
  int* destinations;
  int* sendCounts;
  int* sdispls;
  CG_FLOAT* sendBuffer; /* unused: packing happens in HBM */
  void* dev;            /* sb_halo* */
  CG_UINT* externalGlobal; /* global id of local column nr+i */
} Comm;

/* How ranks talk during SETUP (the reference uses MPI there: Allgather :496,
 * Send/Irecv :134-161).  The launcher provides it: MPI, RCCL (sbh_exchange_rccl),
 * torch.distributed, or nothing for one rank. */
typedef struct {
  void* ctx;
  /* every rank contributes n ints; all receives size*n ints in rank order */
  void (*allgather_ints)(void* ctx, const int* mine, int n, int* all);
  /* rank r gets sendbuf[sdispls[r] .. +sendcounts[r]); counts are already known on
   * both sides */
  void (*alltoallv_ints)(void* ctx, const int* sendbuf, const int* sendcounts,
                         const int* sdispls, int* recvbuf, const int* recvcounts,
                         const int* rdispls);
} sbh_exchange;
void commSetExchange(const sbh_exchange* x); /* NULL: single rank */
const sbh_exchange* sbh_exchange_rccl(void); /* setup exchange over the RCCL communicator */

void commInit(Comm* c, int argc, char** argv);
void commFinalize(Comm* c);
void commDistributeMatrix(Comm* c, MMMatrix* m, MMMatrix* mLocal);
void commPartition(Comm* c, GMatrix* m);
void commPrintConfig(Comm* c, CG_UINT nr, CG_UINT nnz, CG_UINT startRow, CG_UINT stopRow);
/* diagnostics of the VERBOSE build, src/comm.h:54-56 (write to c->logFile when it is open) */
void commGMatrixDump(Comm* c, GMatrix* m);
void commVectorDump(Comm* c, CG_FLOAT* v, CG_UINT size, char* name); /* v: host or device */
void commExchange(Comm* c, CG_UINT numRows, CG_FLOAT* x); /* x: DEVICE vector of nc entries */
void commReduction(CG_FLOAT* v, int op);                   /* v: host or device scalar */
void commPrintBanner(Comm* c);
void commAbort(Comm* c, char* msg);
static inline int commIsMaster(Comm* c) { return c->rank == 0; }
void commBarrier(void);

/* ---- setup: src/matrix.h:51-57 ----------------------------------------------------- */
void MMMatrixRead(MMMatrix* m, char* filename);
void matrixConvertfromMM(MMMatrix* mm, GMatrix* m);
void matrixGenerate(GMatrix* m, Parameter* p, int rank, int size, bool use_7pt_stencil);
/* "irregular": deterministic irregular SPD FE-like matrix of 3*nx*ny*nz rows, the committed stand-in for
 * SuiteSparse Flan_1565 (BASELINE configs[4]; host/sbh_irregular.c).  Enters like a .mtx file: global
 * column ids, rows split over ranks by the file rule (src/comm.c:35-38), b = 1. */
void sbh_matrix_generate_irregular(GMatrix* m, Parameter* p, int rank, int size);
/* binary matrix files, src/matrixBinfile.h:21-22 (same bytes; plain POSIX I/O instead of MPI-IO;
 * SB_BMX_FP64=1 writes the fp64 extension, the reader accepts both) */
typedef struct { /* src/matrixBinfile.h:10-13: one stored nonzero of a .bmx file */
  unsigned int col;
  float val;
} FEntry;
void matrixBinWrite(GMatrix* m, Comm* c, char* filename);
void matrixBinRead(GMatrix* m, Comm* c, char* filename);
/* the driver's matrix set-up, src/main.c:54-84 (generate | generate7P | .mtx | .bmx), and its
 * `-c file.mtx` conversion, src/main.c:41-52 */
void sbh_init_matrix(Comm* c, Parameter* p, GMatrix* m);
/* commDistributeMatrix for a driver in which EVERY rank has read the file: keeps this rank's rows in place */
void sbh_distribute_local(Comm* c, MMMatrix* m, MMMatrix* mLocal);
void sbh_write_bin_matrix(Comm* c, char* mtxFilename);
/* src/allocate.h:9 -- the allocation hook.  With a device up, requests >= 64 KiB come back as memory that lives in HBM, that
 * host loops can store to and that spMVM / waxpby / ddot use in place (src/main.c:205-215 then times the kernel, not staging);
 * see host/sbh_base.c.  sbh_allocate_kind(): what the last request got -- 0 plain host, 1 device-resident and host-visible,
 * 2 pinned host.  sbh_alloc_host(): always plain host memory (what the library's own host-side arrays use). */
void* allocate(size_t alignment, size_t bytesize);
void* sbh_alloc_host(size_t alignment, size_t bytesize);
int sbh_allocate_kind(void);
void sbh_allocate_free(void* p);
double getTimeStamp(void);                          /* src/timing.h:8 */
double getTimeResolution(void);                     /* src/timing.h:9 */

/* runtime-format entry points (what the drop-in symbols below forward to) */
void sbh_convert_crs(CRSMatrix* m, GMatrix* im);
void sbh_convert_scs(SCSMatrix* m, GMatrix* im); /* honours m->C, m->sigma set by the caller */
void sbh_spmv(void* dev_matrix, CG_UINT nr, CG_UINT nc, const CG_FLOAT* x, CG_FLOAT* y);
int sbh_solve_cg(Comm* comm, Parameter* param, void* dev_matrix, CG_UINT nr,
                 const CG_UINT* rowNnz);

/* ---- the hot path: src/solver.h:11-25, src/matrix.h:57 -------------------------------- */
#if defined(CRS) || defined(SCS)
void convertMatrix(Matrix* m, GMatrix* im);
int solveCG(Comm* comm, Parameter* param, Matrix* m);
/* x (nc entries) and y (nr entries) may be device or host pointers; host pointers are
 * staged through HBM (correct, slow: use sb_malloc'ed vectors on the hot path) */
void spMVM(Matrix* m, const CG_FLOAT* restrict x, CG_FLOAT* restrict y);
void commMatrixDump(Comm* c, Matrix* m); /* src/comm.h:55 */
#endif
void waxpby(const CG_UINT n, const CG_FLOAT alpha, const CG_FLOAT* restrict x,
            const CG_FLOAT beta, const CG_FLOAT* restrict y,
            CG_FLOAT* w /* may alias x or y, as src/CGSolver.c:114,127 do (the reference's
                           definition drops `restrict` here too, src/solver.c:21) */);
void ddot(const CG_UINT n, const CG_FLOAT* restrict x, const CG_FLOAT* restrict y,
          CG_FLOAT* restrict result);

/* ---- profiler: src/profiler.h:10-30 --------------------------------------------------- */
typedef enum { WAXPBY = 0, SPMVM, DDOT, COMM, NUMREGIONS } regions;
extern double _t[NUMREGIONS];
/* The reference reads the host clock around a synchronous CPU call (src/profiler.h:18-21).  Here `call` is a stream-ordered
 * launch: the region is bracketed by two device events (sbh_region_begin / _end), nothing waits inside the caller's loop, and
 * profilerPrint() reports the accumulated DEVICE time of every tag that recorded regions (the host-side enqueue time is still
 * added to _t[tag], as the reference does, and is what the table falls back to for a tag without regions). */
#define PROFILE(tag, call)                                                     \
  ts = getTimeStamp();                                                         \
  sbh_region_begin(tag);                                                       \
  call;                                                                        \
  sbh_region_end(tag);                                                         \
  _t[tag] += (getTimeStamp() - ts);
void sbh_region_begin(int tag);
void sbh_region_end(int tag);
void sbh_profile_sync(void); /* waits for the layer's stream (callers that time with the host clock themselves) */
void profilerInit(size_t* facFlops, size_t* facWords);
void profilerPrint(Comm* c, int iterations);
void profilerFinalize(void);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEBENCH_H */
