/* sbhip.h -- the C-ABI of the MI355X (gfx950) HIP layer for SparseBench's CG hot path.
 *
 * Plain C: opaque handles, pointers and sizes only.  This is what a host program
 * written in the reference's language (C) binds; the reference-shaped symbols
 * (convertMatrix, spMVM, waxpby, ddot, solveCG, commExchange, commReduction --
 * see include/sparsebench/) are thin C wrappers over these entry points, and
 * INTEGRATION.md shows the binding a SparseBench maintainer would add.
 *
 * Conventions (mirroring the reference):
 *  - one process drives one GPU; all calls come from that process's main thread
 *    (reference: src/profiler.c:17 globals, single-threaded API use);
 *  - errors are fatal: message with file:line on stderr, then exit(EXIT_FAILURE)
 *    (reference: src/allocate.c:19-33, src/matrix.c:129-170) -- no error codes;
 *  - CG_FLOAT = double, CG_UINT = unsigned int (reference defaults,
 *    src/util.h:35-53, config.mk:7-8);
 *  - every vector pointer is a DEVICE pointer unless the name says host;
 *  - work is enqueued on the layer's own HIP stream; calls that return a value to
 *    the host synchronise that stream, the others do not.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the reference root).
 */
#ifndef SBHIP_H
#define SBHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sb_matrix sb_matrix; /* device-resident sparse matrix (CRS or SCS) */
typedef struct sb_halo sb_halo;     /* device-resident halo plan of this rank     */
typedef struct sb_cg sb_cg;         /* CG solver state (vectors + scalars in HBM) */

/* ---- context ------------------------------------------------------------- */
/* replaces: commInit's process setup, src/comm.c:863-878 (device instead of rank) */
void sb_init(int device);
void sb_finalize(void);
int sb_is_initialized(void);
int sb_device_count(void);
const char* sb_device_name(void); /* e.g. "AMD Instinct MI355X (gfx950)" */
int sb_num_cus(void);
void sb_sync(void);      /* wait for the layer's stream */
void* sb_stream(void);   /* the hipStream_t, for callers that want to order work */

/* replaces: allocate(), src/allocate.h:9 -- for arrays that live on the hot path */
void* sb_malloc(size_t bytes);
void sb_free(void* dev);
void sb_memset(void* dev, int byte, size_t bytes);
void sb_h2d(void* dev, const void* host, size_t bytes); /* synchronous */
void sb_d2h(void* host, const void* dev, size_t bytes); /* synchronous */
void sb_d2d(void* dst, const void* src, size_t bytes);  /* stream-ordered */
int sb_is_device_ptr(const void* p);
/* replaces: allocate() AS THE DRIVER USES IT, src/allocate.c:12-36 + src/main.c:205-211 -- vectors the caller fills with host
 * loops and then hands to spMVM.  Memory in HBM that the CPU can store to (fine-grained device memory through the PCIe BAR) and
 * kernels read in place; NULL where the one-time probe says this box cannot do that (sb_host_visible_reason() tells why either
 * way; SPARSEBENCH_ALLOCATE=host switches it off).  sb_is_device_ptr() is 1 for it; free with sb_free(). */
void* sb_malloc_host_visible(size_t bytes);
const char* sb_host_visible_reason(void);
void* sb_malloc_pinned_host(size_t bytes); /* fallback: pinned host memory (staged by spMVM / waxpby / ddot); NULL on failure */
void sb_free_pinned_host(void* p);
/* host <-> device copies made through sb_h2d / sb_d2h so far: {h2d calls, h2d bytes, d2h calls, d2h bytes} */
void sb_copy_counters(uint64_t out[4]);

/* stream-ordered timing (PROFILE macro, src/profiler.h:18-21, needs completion) */
void* sb_event_create(void);
void sb_event_record(void* ev);
float sb_event_elapsed_ms(void* start, void* stop); /* synchronises on stop */
void sb_event_destroy(void* ev);
/* replaces: the PROFILE macro's host clock, src/profiler.h:18-21 -- a region (tag 0..7) bracketed by two events on the layer's
 * stream, nothing waits inside the caller's loop; sb_region_seconds() waits for what is outstanding and returns the accumulated
 * DEVICE time of the tag and how many regions were recorded. */
void sb_region_begin(int tag);
void sb_region_end(int tag);
double sb_region_seconds(int tag, uint64_t* count);
void sb_region_reset(void);

/* ---- matrix upload (device side of convertMatrix, src/matrix.h:57) -------- */
/* CRS: src/CRSMatrix.h:9-16.  Arrays are host pointers, copied to HBM. */
sb_matrix* sb_crs_upload(uint32_t nr, uint32_t nc, const uint32_t* rowPtr,
                         const uint32_t* colInd, const double* val);
/* Sell-C-sigma: src/SCSMatrix.h:13-27, host layout exactly as the reference's
 * convertMatrix builds it (src/matrix-SCS.c:31-196; columns NOT permuted).
 * For sigma > 1 the device copy of colInd is renumbered to the permuted row order
 * (symmetric permutation) so the CG vectors can live in permuted order. */
sb_matrix* sb_scs_upload(uint32_t nr, uint32_t nc, uint32_t C, uint32_t sigma,
                         uint32_t nChunks, uint32_t nElems, const uint32_t* chunkPtr,
                         const uint32_t* chunkLens, const uint32_t* colInd, const double* val,
                         const uint32_t* oldToNewPerm, const uint32_t* newToOldPerm);
void sb_matrix_free(sb_matrix* m);
/* Where the reference-layout stream (val, colInd: what spMVM of src/matrix-SCS.c:198-228 / src/matrix-CRS.c:46-65 reads) sits
 * in device memory: one slab, the two arrays at the given offsets (0..256 MB each) inside their regions of it.  The same kernel
 * streams the same bytes 10-20 % faster or slower depending on WHICH memory that is; sb_scs_upload / sb_crs_upload
 * pick it by measurement, together with the memory of the CG loop's vectors (SB_PLACE=0: everything stays where hipMalloc put
 * it).  The calls below (down to sb_placement_probe) are lab calls for uploads made with SB_PLACE=0 (tools/placement_lab*.py). */
void sb_matrix_place(sb_matrix* m, int colOffMB, int valOffMB);
void sb_matrix_place_fresh(sb_matrix* m); /* the same arrays in a NEW slab; earlier slabs stay allocated until the commit */
void sb_matrix_place_commit(sb_matrix* m);
void sb_matrix_placement(const sb_matrix* m, int out[2]); /* {-1, -1}: not placed */
void sb_matrix_place_at(sb_matrix* m, void* colMem, void* valMem); /* lab: the stream copied to the caller's memory */
void sb_matrix_place_home(sb_matrix* m);                            /* lab: back to the first upload */
size_t sb_placement_arena_bytes(const sb_matrix* m);                /* lab: bytes of the loop's vector layout */
float sb_placement_probe(sb_matrix* m, void* arena);                /* lab: the tuner's proxy step (us) with the vectors at `arena` */
int sb_matrix_placement_report(const sb_matrix* m, float us[3]); /* probes the upload's tuner timed; us of a proxy loop body: first arena + hipMalloc's placement, the pair kept, the slowest */
void sb_matrix_debug_ptrs(const sb_matrix* m, unsigned long long out[4]); /* lab: device addresses of colInd, val, chunkPtr | rowPtr, chunkLens */
uint32_t sb_matrix_nr(const sb_matrix* m);
uint32_t sb_matrix_nc(const sb_matrix* m);
int sb_matrix_is_permuted(const sb_matrix* m); /* 1 for SCS with a non-identity row sort */
/* algorithmic bytes one SpMV moves:
 *   CRS 12*nnz + 4*(nr+1) + 8*nr + 8*nc ; SCS 12*nElems + 8*nChunks + 8*nrPadded + 8*nc */
double sb_matrix_spmv_bytes(const sb_matrix* m);
/* SCS C=64 matrices also get a device-private LOSSLESS compressed mirror at upload
 * (csrc/pack.hip.h; SB_PACK=0 disables it): level 1 = 16-bit column offsets per chunk,
 * level 2 = additionally a <=256-entry value dictionary.  Results are bit-identical to
 * the uncompressed kernel; the host-visible arrays keep the reference layout. */
/* Optional, before an upload of a rank-local matrix: the global ids of its halo columns (local column nr + i has
 * global id global_ids[i]; commPartition's Comm::externalGlobal).  Used only to lay out device-private x windows
 * (the partitioner numbers halo columns in the order it met them; in ascending global order a stencil's halo
 * plane looks like the rank's own planes and the pattern levels apply to the tiles next to a rank boundary).
 * Stays in force until replaced (n = 0 clears); uploads whose nc - nr differs from n ignore it. */
void sb_set_external_ids(const uint32_t* global_ids, uint32_t n);
int sb_matrix_pack_level(const sb_matrix* m);
/* select the SpMV kernel at run time.  The product has two: 0 the reference-layout stream, 5 the masked row programs
 * (level 6) where the matrix has them (the default then); any other request falls to the next lower of the two.
 * Lab builds (sb_lab_build) also run the intermediate levels as kernels of their own: 1 packed stream with x
 * gathered through the cache, 2 packed stream with each workgroup's x window staged in LDS
 * (built when every tile's window fits; SB_PACK=2 stops at mode 1), 3 one-byte pattern
 * codes naming (value, window-slot delta) pairs + LDS window (built when every tile has
 * <= 255 distinct pairs; SB_PACK=3 stops at mode 2), 5 the same tiles with every chunk stored as a
 * masked row program (built when >= 98 % of the chunks' rows are sub-sequences of a row of their tile;
 * SB_PACK=5 stops at mode 3; there is no mode 4).  Clamped to what the matrix has: a mode it lacks
 * falls to the next lower one.  Default = 5 where built, else 3 for matrices of more than one round of
 * resident workgroups (8 tiles of 256 rows per CU) and 2 below that.  All modes give bit-identical results. */
/* CRS matrices: 0 = native CRS kernel, 3 / 5 = product through a device-private Sell-64-1 pattern mirror whose
 * padding is not added (exactly the CRS loop's sums); built when the matrix has repeating row patterns. */
void sb_matrix_use_packed(sb_matrix* m, int mode);
int sb_matrix_packed_mode(const sb_matrix* m);
/* which native CRS kernel streams the reference's arrays (pack mode 0): 1 spmv_crs_split -- equal windows of nonzeros, the
 * default where no row is longer than 1025 --, 0 spmv_crs_stream (row blocks; SB_CRS_KERNEL=stream forces it).  Both:
 * src/matrix-CRS.c:46-65, same bits. */
int sb_matrix_crs_kernel(const sb_matrix* m);
uint32_t sb_matrix_lds_window(const sb_matrix* m); /* doubles per workgroup, 0 if not built */
uint32_t sb_matrix_pattern_classes(const sb_matrix* m); /* pattern tables built (mode 3), 0 if none */
/* mode 3, level 5: distinct shared row patterns; *uniformChunks = chunks stored as one row
 * pattern + exception lanes (the rest keep per-lane codes).  SB_PACK=4 builds none. */
uint32_t sb_matrix_row_patterns(const sb_matrix* m, uint32_t* uniformChunks);
/* mode 5, level 6: distinct masked row programs; *maskedChunks = chunks stored as one program + per-row base slots
 * (every lane runs the program, each add under the mask of the lanes that have the entry).  0 if not built
 * (SB_PACK=5 builds none; fewer than 98 % of the chunks qualifying: not kept). */
uint32_t sb_matrix_row_programs(const sb_matrix* m, uint32_t* maskedChunks);
/* bytes the selected SpMV kernel really moves per launch (stream + x + y) */
double sb_matrix_stream_bytes(const sb_matrix* m);

/* ---- kernels ---------------------------------------------------------------- */
/* spMVM, src/solver.h:13: y = A x, x has nc entries, y has nr entries, both in
 * the caller's (original) row order for every format and sigma. */
void sb_spmv(const sb_matrix* m, const double* x, double* y);
/* the SCS fast path used inside CG: x and y in the matrix's permuted row order
 * (identical to sb_spmv when sb_matrix_is_permuted() == 0) */
void sb_spmv_native(const sb_matrix* m, const double* x, double* y);
/* sb_spmv_native plus, fused into the same launch, the partial sums of the dot product x . y in the canonical order of
 * DESIGN 4.3 (rows of the device's order) -- what the CG loop does for p . Ap.  Returns which values partials_dev holds:
 *   0  nothing: the selected kernel has no fused dot (Sell-C-sigma with C != 64, native CRS);
 *   2  LEVEL-1 values, one per aligned 256 rows = ((q0 + q1) + q2) + q3 of four level-0 partials: ceil(nr / 256) doubles
 *      (the product's kernels: reference-layout Sell-64 stream and masked row programs -- a block / tile combines its
 *      chunks itself, so the scalar step that follows reads a quarter of the bytes through its single CU);
 *   1  level-0 partials, one per 64 rows: 4 * ceil(nr / 256) doubles (lab-only kernels).
 * partials_dev: 4 * ceil(nr / 256) doubles, zero-filled by the caller (entries behind the last group stay +0.0). */
int sb_spmv_native_dot(const sb_matrix* m, const double* x, double* y, double* partials_dev);
/* vector <-> permuted order of an SCS matrix: out[new] = in[old] / out[old] = in[new] */
void sb_permute(const sb_matrix* m, const double* in_orig, double* out_perm);
void sb_unpermute(const sb_matrix* m, const double* in_perm, double* out_orig);

/* waxpby, src/solver.h:15-20: w = alpha*x + beta*y (w may alias x or y) */
void sb_waxpby(uint32_t n, double alpha, const double* x, double beta, const double* y,
               double* w);
/* ddot, src/solver.h:22-25 (+ commReduction SUM when a communicator is attached).
 * Fixed summation order (DESIGN.md "dot order"): bit-reproducible run to run. */
void sb_ddot_async(uint32_t n, const double* x, const double* y, double* result_dev);
double sb_ddot(uint32_t n, const double* x, const double* y); /* synchronises */
/* the stages of the fixed order, exposed for parity tests: level 0 writes one partial per
 * 64 elements into partials_dev[0 .. 4*ceil(n/256)) (tail zeroed); sb_reduce_final does
 * levels 1-2 over m = ceil(n/256) groups of four partials */
void sb_ddot_partials(uint32_t n, const double* x, const double* y, double* partials_dev);
void sb_reduce_final(uint32_t m, const double* partials_dev, double* result_dev);

/* ---- multi-GPU (one rank per GPU, RCCL over xGMI) ---------------------------- */
/* replaces MPI_Init / MPI_COMM_WORLD.  id = 128-byte ncclUniqueId made by rank 0
 * with sb_comm_unique_id() and handed to the other ranks by the launcher. */
#define SB_UNIQUE_ID_BYTES 128
void sb_comm_unique_id(void* id_out);
void sb_comm_init(int rank, int size, const void* id);
/* Alternative to RCCL: a host-mediated transport supplied by the launcher (MPI without
 * GPU awareness, gloo, ...).  Also what lets the N-rank device path be exercised by several
 * processes sharing ONE GPU (tests/test_gpu_multirank.py).  Callbacks are entered with the
 * layer's stream synchronised, get DEVICE pointers, and return when those are written. */
typedef struct {
  void* ctx;
  /* commReduction: in place on one double; op 0 = MAX, 1 = SUM */
  void (*allreduce)(void* ctx, double* v_dev, int op);
  /* commExchange after packing: send_dev[sdispls[i] .. +sendCounts[i]) goes to destinations[i];
   * recv_dev[rdispls[j] .. +recvCounts[j]) comes from sources[j] */
  void (*neighbour_exchange)(void* ctx, const double* send_dev, int outdegree, const int* destinations,
                             const int* sendCounts, const int* sdispls, double* recv_dev, int indegree,
                             const int* sources, const int* recvCounts, const int* rdispls);
  /* optional (may be NULL): all_host[r * nbytes ..] = rank r's mine_host; lets the layer set up the
   * exchange over peer-mapped memory (halo staging areas) on top of this transport */
  void (*allgather_bytes)(void* ctx, const void* mine_host, int nbytes, void* all_host);
} sb_transport;
void sb_comm_init_transport(int rank, int size, const sb_transport* t);
/* In-kernel all-reduce of the CG scalars over peer-mapped memory (one launch per dot instead of
 * local reduce | ncclAllReduce | scalar step).  Every rank owns a small fine-grained buffer that
 * its peers map through HIP IPC.  sb_comm_init does this by itself (the handles travel over RCCL).
 * A launcher with its own transport calls sb_comm_p2p_handle on every rank, gathers the
 * SB_P2P_HANDLE_BYTES of all ranks in rank order, and calls sb_comm_p2p_open (collective): peers are
 * mapped, one exchange is tested, and the ranks agree over the transport; any failure on any rank
 * leaves every rank on the transport's all-reduce.  SB_P2P=0 disables.  Returns 1 when enabled. */
#define SB_P2P_HANDLE_BYTES 64
int sb_comm_p2p_handle(unsigned char* handle_out);
int sb_comm_p2p_open(const unsigned char* all_handles); /* NULL: this rank has no handle */
int sb_comm_p2p_enabled(void);
/* one line saying why the path is on or off (which rank, which call failed, what the self-test saw);
 * waits inside CG are bounded by SB_P2P_TIMEOUT_MS (default 30000), the set-up self-tests by 5 s */
const char* sb_comm_p2p_reason(void);
/* Which data plane the CG loop uses from now on: 1 (default) the peer-mapped paths where their set-up succeeded,
 * 0 the communicator's own collectives (all-reduce, send / recv: src/comm.c:640-648,659) although the mappings
 * exist -- as SB_P2P=0 SB_P2P_HALO=0 would have given, nothing torn down, so one process can time both (bench.py).
 * Collective by contract: every rank calls it with the same value, between solves; solver objects are created
 * after the switch. */
void sb_comm_data_plane(int peer_mapped);
int sb_comm_data_plane_selected(void);
/* Variant of the peer-mapped halo exchange: 1 = the halo push rides in the SpMV launch (its first workgroups send
 * p[elementsToSend], src/comm.c:635-638) instead of a launch of its own; 0 = separate push kernel (default;
 * SB_HALO_PUSH_INSIDE=1 changes the default).  Same bits.  Collective, between solves. */
void sb_comm_halo_push_inside(int on);
/* what the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice):
 * out = {ranks, this rank, HIP device}; returns 0 (out = -1) without an RCCL communicator */
int sb_comm_rccl_info(int out[3]);
void sb_comm_finalize(void);
int sb_comm_rank(void);
int sb_comm_size(void);
/* commReduction, src/comm.h:58 (op: 0 = MAX, 1 = SUM as enum op, src/comm.h:25);
 * in place on one device double; stream-ordered */
void sb_comm_reduction(double* v_dev, int op);
/* setup-time exchanges between ranks over the same communicator, host buffers in
 * and out (replace MPI_Allgather src/comm.c:496 and the Send/Irecv of wanted ids
 * src/comm.c:134-161) */
void sb_comm_allgather_bytes(const void* mine_host, int nbytes, void* all_host);
void sb_comm_alltoallv_ints(const int* sendbuf, const int* sendcounts, const int* sdispls,
                            int* recvbuf, const int* recvcounts, const int* rdispls);
void sb_comm_barrier(void);
/* device side of commPartition's result, src/comm.h:27-46: neighbour lists and
 * elementsToSend become device arrays.  perm_of_row may be NULL (identity). */
/* (test hook: SB_TEST_CORRUPT_HALO=r makes rank r send a wrong row's value in its first halo slot, announced on
 * stderr -- bench.py's pre-flight gate must catch it) */
sb_halo* sb_halo_create(uint32_t nr, int outdegree, const int* destinations,
                        const int* sendCounts, const int* sdispls, int indegree,
                        const int* sources, const int* recvCounts, const int* rdispls,
                        const int* elementsToSend, int totalSendCount, int externalCount,
                        const uint32_t* oldToNewPerm);
void sb_halo_free(sb_halo* h);
/* 1: inside CG this halo is exchanged by push / pull kernels over peer-mapped staging areas (set up
 * collectively in sb_halo_create when the in-kernel all-reduce is on; SB_P2P_HALO=0 disables), 0: RCCL /
 * transport send-recv */
int sb_halo_p2p_enabled(const sb_halo* h);
const char* sb_halo_p2p_reason(const sb_halo* h);
/* commExchange, src/comm.h:57: pack x[elementsToSend] and deliver every
 * neighbour's slice into x[numRows ...]; stream-ordered */
void sb_halo_exchange(sb_halo* h, double* x);

/* ---- CG (solveCG, src/solver.h:11, src/CGSolver.c:62-141) --------------------- */
/* b_host / xexact_host: nr doubles in original row order (xexact may be NULL). */
sb_cg* sb_cg_create(const sb_matrix* m, sb_halo* halo, const double* b_host,
                    const double* xexact_host);
void sb_cg_free(sb_cg* s);
void sb_cg_debug_ptrs(const sb_cg* s, unsigned long long out[8]); /* lab: device addresses of r, p, p', Ap, x, b, partials, control block */
/* fused = 0: the reference's op list (waxpby, spMVM, ddot as separate launches); 1 (default): dots fused into the
 * SpMV / update kernels (5 launches per loop body); 2: additionally the vector phase of a body
 * (alpha | x, r update + r.r | beta, loop test | p update; src/CGSolver.c:124-128 and :107-116) as ONE launch
 * whose workgroups wait for each other -- used when this rank has its GPU to itself (SB_SHARED_GPU=1 says it
 * has not), the all-reduce is the in-kernel one (or there is one rank) and the rows fit the resident grid's
 * registers; otherwise it behaves as 1; 3: the two scalar steps ride in front of their consumers (workgroup 0 of the
 * r / p update takes them and publishes alpha / beta through a flag; 3 launches per body; one rank only, otherwise as 1).  (2 and 3 measured slower than 1 at 128^3: sbhip_cg.inc.h.)  Same bits in every mode. */
void sb_cg_set_fused(sb_cg* s, int fused);
/* spans per wave of the one-launch vector phase the solver will use, 0 if it will not use it */
int sb_cg_vector_phase(sb_cg* s);
/* launches per loop body the loop will use: 5 (p update | SpMV | alpha | r update | beta), 3 (fused = 3), 2 (fused = 2);
 * 0 for the reference's op list */
int sb_cg_launches_per_body(sb_cg* s);
/* several ranks: the count above includes the halo kernels (peer-mapped push: +1, or +0 riding in the SpMV launch; pack
 * kernel in front of a send / recv group: +1) and, without the in-kernel all-reduce, one more kernel per dot (+2);
 * the communicator calls themselves (2 all-reduces, 1 send-recv group) are counted here: 0 on the peer-mapped paths */
int sb_cg_collectives_per_body(sb_cg* s);
/* The p update INSIDE the SpMV (round 3; pack.hip.h: spmv_prog_fusep): p = r + beta p (src/CGSolver.c:114; k = 1: p = r,
 * :109), the x update the previous body owes (:127) and Ap = A p with its p.Ap values (:123-125) as ONE launch -- every tile
 * forms p_new for its x window while it stages it and stores p_new / x for its own rows; p is double-buffered; on several
 * ranks the halo push forms the boundary values itself.  4 launches per body instead of 5; element for element the same
 * arithmetic in the same order: same bits.  Used where every chunk of the matrix is a masked row program with a mapped or
 * simple window, in the default loop (fused = 1), on one rank or with the peer-mapped halo; otherwise the separate p update.
 * on = 1 / 0 selects / deselects it, -1 = default (SB_FUSE_P, else the library's choice).  sb_cg_fuse_p: what the loop will do. */
void sb_cg_set_fuse_p(sb_cg* s, int on);
int sb_cg_fuse_p(sb_cg* s);
/* The alpha step (src/CGSolver.c:124-126) inside the r update's launch: every workgroup of the r update reduces the p.Ap
 * values itself in the canonical order (identical bits everywhere), workgroup 0 records the step -- one launch fewer per loop
 * body on one rank (sb_cg_launches_per_body tells).  on = 1 / 0, -1 = default (SB_FUSE_ALPHA, else on).  Same bits. */
void sb_cg_set_fuse_alpha(sb_cg* s, int on);
/* The beta step / loop test (src/CGSolver.c:107, :111-113, :116) at the head of the next body's p update, where that is a launch
 * of its own (not inside the SpMV): taken by every workgroup, recorded by workgroup 0.  One rank: each workgroup reduces the r.r
 * values itself; several ranks on the communicator's collectives: the all-reduced sum is read, the third launch of the dot goes
 * (as for alpha: sb_cg_set_fuse_alpha).  Every sb_cg_run_iters call still leaves the loop state complete (a step left owing at
 * its end is taken by a launch of its own).  on = 1 / 0, -1 = default (SB_FUSE_BETA, else on).  Same bits. */
void sb_cg_set_fuse_beta(sb_cg* s, int on);
void sb_cg_set_graph(sb_cg* s, int use_graph);
/* Runs solveCG's whole loop without host synchronisation; returns k exactly as
 * the reference does (src/CGSolver.c:140).  Blocking. */
int sb_cg_solve(sb_cg* s, int itermax, double eps);
/* The same in three steps, for callers that time a slice of the loop (bench.py):
 *   sb_cg_start      x0 = 0, prologue (src/CGSolver.c:94-103), loop test for k = 1; enqueues only
 *   sb_cg_run_iters  enqueue the next `iters` loop bodies (k = 1, 2, ...); never touches the host;
 *                    bodies past the reference's loop exit (k >= itermax or normr <= eps) are no-ops
 *   sb_cg_finish     wait and return k as solveCG does                                        */
void sb_cg_start(sb_cg* s, int itermax, double eps);
void sb_cg_run_iters(sb_cg* s, int iters);
int sb_cg_finish(sb_cg* s);
/* every r.r (index 0 = prologue) and p.Ap the solve produced, full precision */
int sb_cg_history(const sb_cg* s, double* rr_out, int rr_cap, double* pAp_out, int pAp_cap,
                  int* n_pAp);
void sb_cg_solution(const sb_cg* s, double* x_host); /* original row order */
double sb_cg_check_residual(const sb_cg* s);         /* max|x-xexact|, src/CGSolver.c:40-60 */
/* per-region milliseconds of the last unfused solve: [waxpby, spMVM, ddot, comm]
 * (regions of src/profiler.h:24) */
void sb_cg_region_ms(const sb_cg* s, double out[4]);
/* milliseconds the last solve's loop (k = 1 .. itermax-1) took on the GPU: what the
 * reference brackets with timeStart/timeStop (src/CGSolver.c:106,130) */
double sb_cg_loop_ms(const sb_cg* s);

/* bench instrumentation: bracket every SpMV launch of the loop with HIP events on the
 * layer's stream; sb_cg_spmv_ms returns their summed duration and count */
void sb_cg_spmv_timing(sb_cg* s, int on);
double sb_cg_spmv_ms(sb_cg* s, int* launches);
/* lab call: the same launches one by one (microseconds each); returns their number, writes at most cap */
int sb_cg_spmv_us_series(sb_cg* s, float* out, int cap);
/* per-kernel breakdown of the loop: an event after every launch of a loop body.  sb_cg_phase_ms returns the number
 * of phases P <= 8 and fills ms_out[i] / count_out[i] (summed milliseconds, occurrences) since timing was switched on:
 * 0 p update (+ owed x update), 1 halo (push kernel, or pack + send/recv), 2 SpMV (+ fused p.Ap partials; with the
 * peer-mapped halo: incl. the wait for the neighbours' pushes), 3 alpha step (levels 1-2 of p.Ap [+ all-reduce] +
 * alpha), 4 r update (+ r.r partials), 5 beta step / loop test (levels 1-2 of r.r [+ all-reduce] + beta),
 * 6 separate dot passes (reference op list, native CRS).  Every event costs ~1 us: never on in a clean timing. */
void sb_cg_phase_timing(sb_cg* s, int on);
int sb_cg_phase_ms(sb_cg* s, double ms_out[8], int count_out[8]);
/* device control block: out = {stop, stop_next, iters, n_rr, n_pAp} (proof that the
 * timed iterations really ran) */
void sb_cg_counters(const sb_cg* s, int out[5]);

/* debug/measurement: raw streaming-read rate of the device in GB/s (DESIGN.md uses it
 * as the measured ceiling next to the 8 TB/s spec) */
double sb_debug_stream_read_gbs(size_t bytes, int reps);

const char* sb_version(void);
/* 1: a lab build (-DSB_LAB, `make lab`): the product plus the alternatives that were measured slower and are kept for
 * the record -- compressed-mirror levels 1-5 as SpMV kernels of their own (sb_matrix_use_packed 1-3),
 * sb_cg_set_fused 2 / 3, sb_cg_set_graph, SB_HALO_OVERLAP.  0: the product, in which those requests fall back to
 * what ships (modes 0 / 5, fused 0 / 1, no graph). */
int sb_lab_build(void);

#ifdef __cplusplus
}
#endif
#endif /* SBHIP_H */
