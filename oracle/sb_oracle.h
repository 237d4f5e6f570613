/* sb_oracle.h -- CPU oracle for the SparseBench CG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm for the hot path (generator -> partition -> CRS / Sell-C-sigma
 * conversion -> SpMV, waxpby, ddot, CG).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (sparsebench_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * Parity pinning: checked bit-for-bit against (a) the reference's own golden
 * fixtures tests/data/expected/ *.in (copied as data into tests/golden/ref/),
 * (b) the reference sources compiled strict-IEEE where they lie
 * (oracle/_ref/, built by oracle/build_ref.sh) and (c) committed residual
 * histories captured from that build (the .json files in tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#ifndef SB_ORACLE_H
#define SB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* General (format-neutral) local matrix: src/matrix.h:29-35 (GMatrix), stored
 * SoA instead of the reference's AoS Entry{col,val}. */
typedef struct {
  uint32_t nr, nc;
  uint32_t nnz;      /* as the reference reports it (27*nr for generated) */
  uint32_t nnzTrue;  /* rowPtr[nr] */
  uint32_t totalNr, totalNnz, startRow, stopRow;
  int generated;     /* 1: HPCG generator, b = 27-(nnzrow-1); 0: file, b = 1 */
  uint32_t* rowPtr;  /* nr+1 */
  uint32_t* col;     /* nnzTrue */
  double* val;       /* nnzTrue */
} orc_gmatrix;

/* Halo plan of one rank: src/comm.h:27-46 (Comm, _MPI part). */
typedef struct {
  int rank, size;
  int externalCount, totalSendCount;
  int indegree, outdegree;
  int* sources;      /* indegree, ascending */
  int* recvCounts;
  int* rdispls;
  int* destinations; /* outdegree, ascending */
  int* sendCounts;
  int* sdispls;
  int* elementsToSend; /* totalSendCount local row ids */
  uint32_t* externalGlobal; /* externalCount: global id of local col nr+i */
} orc_plan;

/* Sell-C-sigma: src/SCSMatrix.h:13-27 */
typedef struct {
  uint32_t nr, nc, nnz, C, sigma, nrPadded, nChunks, nElems;
  uint32_t* chunkPtr;  /* nChunks+1 */
  uint32_t* chunkLens; /* nChunks */
  uint32_t* colInd;    /* nElems */
  double* val;         /* nElems */
  uint32_t* oldToNewPerm; /* nr */
  uint32_t* newToOldPerm; /* nr */
} orc_scs;

/* ---- setup ---------------------------------------------------------- */
orc_gmatrix* orc_generate(int nx, int ny, int nz, int rank, int size, int use7pt);
orc_gmatrix* orc_mm_load(const char* path);           /* whole file, rank 0 of 1 */
orc_gmatrix* orc_mm_load_part(const char* path, int rank, int size);
orc_gmatrix* orc_gm_from_arrays(uint32_t nr, uint32_t nc, const uint32_t* rowPtr,
                                const uint32_t* col, const double* val);
void orc_gm_free(orc_gmatrix* g);

/* Partition P local matrices (global column ids) in place; returns P plans. */
orc_plan* orc_partition(orc_gmatrix** locals, int P);
void orc_plan_free(orc_plan* plans, int P);

orc_scs* orc_convert_scs(const orc_gmatrix* g, uint32_t C, uint32_t sigma);
void orc_scs_free(orc_scs* s);

/* ---- kernels -------------------------------------------------------- */
void orc_spmv_crs(const orc_gmatrix* g, const double* x, double* y);
/* fixed semantics: y has nr entries in ORIGINAL row order */
void orc_spmv_scs(const orc_scs* s, const double* x, double* y);
/* literal reference semantics: y has nrPadded entries in PERMUTED order */
void orc_spmv_scs_literal(const orc_scs* s, const double* x, double* y);
void orc_waxpby(uint32_t n, double alpha, const double* x, double beta,
                const double* y, double* w);
double orc_ddot_seq(uint32_t n, const double* x, const double* y);
double orc_ddot_tree(uint32_t n, const double* x, const double* y);
/* Dot2 (twice the working precision, rounded once): the "exactly rounded" yardstick; orc_cg dot_mode 2 */
double orc_ddot_exact(uint32_t n, const double* x, const double* y);
/* canonical order pieces, exposed so tests can check each GPU stage */
void orc_ddot_partials(uint32_t n, const double* x, const double* y, double* partials);
double orc_reduce_final(uint32_t m, const double* partials);

/* ---- CG -------------------------------------------------------------
 * fmt 0 = CRS, 1 = SCS(C,sigma).  dot_mode 0 = sequential (reference),
 * 1 = canonical tree (what the HIP kernels do).  rank_sum 0 = ranks summed
 * 0..P-1 in order, 1 = pairwise tree (recursive doubling).
 * rr[0..] receives every r.r (rr[0] = prologue), pAp[0..] every p.Ap.
 * Returns k as solveCG does (src/CGSolver.c:140).                       */
int orc_cg(orc_gmatrix** locals, const orc_plan* plans, int P, int fmt, uint32_t C,
           uint32_t sigma, int itermax, double eps, int dot_mode, int rank_sum,
           double* rr, int* n_rr, double* pAp, int* n_pAp, double** x_out,
           double* max_err);

/* timing helper for bench.py's cpu_baseline leg (OpenMP if built with it) */
double orc_time_cg_iters(orc_gmatrix* g, int iters, int* threads_used);
double orc_time_spmv(orc_gmatrix* g, int reps, int* threads_used);

#ifdef __cplusplus
}
#endif
#endif
