// sbhip_cg.inc.h -- part of the single translation unit sbhip.hip (textual include, shares its
// static context): the device-resident CG loop.
// ===========================================================================
// CG
// ===========================================================================
enum { R_WAXPBY = 0, R_SPMVM = 1, R_DDOT = 2, R_COMM = 3 };

// events that time kernels inside the loop: created without the system-scope fence a default event carries (SB_EVENT_FLAGS
// overrides the creation flags; 0 = default events)
static unsigned timing_event_flags()
{
  static const unsigned f = getenv("SB_EVENT_FLAGS") ? (unsigned)strtoul(getenv("SB_EVENT_FLAGS"), nullptr, 0) : (unsigned)hipEventDisableSystemFence;
  return f;
}
static void mark(sb_cg* s, int region)
{ // region = the region that ENDS here (-1: start marker)
  if (!s->timing) return;
  if (s->evUsed == s->ev.size()) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    s->ev.push_back(e);
    s->evRegion.push_back(-1);
  }
  s->evRegion[s->evUsed] = region;
  HIP_CHECK(hipEventRecord(s->ev[s->evUsed], g.stream));
  s->evUsed++;
}

// Per-kernel breakdown of the loop (bench.py): with sb_cg_phase_timing on, an event follows every launch of a loop
// body; the time between two consecutive events belongs to the phase that ENDS at the second one.  Events serialise
// nothing the stream does not already serialise, but each costs ~1 us of its own: the clean it/s is never taken with
// them on.
enum { PH_P_UPDATE = 0, PH_HALO = 1, PH_SPMV = 2, PH_ALPHA = 3, PH_R_UPDATE = 4, PH_BETA = 5, PH_DOT_PASS = 6, PH_COUNT = 7 };
static void phase_mark(sb_cg* s, int ph)
{ // ph = the phase that ends here (-1: start marker)
  if (!s->phaseTiming) return;
  if (s->phUsed == s->phEv.size()) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, timing_event_flags()));
    s->phEv.push_back(e);
    s->phId.push_back(-1);
  }
  s->phId[s->phUsed] = ph;
  HIP_CHECK(hipEventRecord(s->phEv[s->phUsed], g.stream));
  s->phUsed++;
}

// ---- where the loop's vectors live ------------------------------------------------------------------------------------------
// r, Ap, x, b, p, p' (and xexact) sit in ONE allocation, laid out by vec_layout (sbhip_launch.inc.h).  Which memory that is moves
// the CG step of the section-8d loop by up to 10 %: the upload's placement tuner has measured a good one together with the
// matrix's own placement (sb_matrix::vecArena) and the first sb_cg of a matrix takes it; a second sb_cg on the same matrix while
// the first is alive allocates its own.
static void cg_point_vectors(sb_cg* s, char* slab, bool hasExact)
{
  const VecLayout L = vec_layout(s->nr, s->nc, hasExact, s->vecPad);
  auto at = [&](size_t o) { return reinterpret_cast<double*>(slab + o); };
  s->vecSlab = slab;
  s->r = at(L.r), s->Ap = at(L.Ap), s->x = at(L.x), s->b = at(L.b);
  s->p = at(L.p); // nc = nr + externals (src/CGSolver.c:70)
  s->pbuf[0] = s->p;
  s->pbuf[1] = at(L.p2); // second p of the fused p update (fusep_plan)
  s->xexact  = hasExact ? at(L.xexact) : nullptr;
}

sb_cg* sb_cg_create(const sb_matrix* m, sb_halo* halo, const double* b_host, const double* xexact_host)
{
  need_init();
  sb_cg* s = new sb_cg();
  s->A = m, s->halo = halo, s->nr = m->nr, s->nc = m->nc;
  if (halo && halo->nr != m->nr) SB_FATAL("halo plan and matrix disagree on nr");
  if (halo && m->nr + (uint32_t)halo->externalCount != m->nc) SB_FATAL("halo externalCount != nc-nr");
  const size_t nb = (size_t)m->nr * sizeof(double);
  // r, Ap, x, b, p, p' (and xexact) in ONE allocation: the matrix's tuned arena if it is free, else an allocation of their own
  s->vecPad       = getenv("SB_CG_VEC_PAD_KB") ? (size_t)atol(getenv("SB_CG_VEC_PAD_KB")) << 10 : 0;
  s->vecSlabBytes = vec_layout(s->nr, s->nc, xexact_host != nullptr, s->vecPad).total;
  {
    sb_matrix* A = const_cast<sb_matrix*>(m);
    if (A->vecArena && !A->vecArenaBusy && s->vecPad == 0 && s->vecSlabBytes <= A->vecArenaBytes) {
      A->vecArenaBusy = true, s->vecFromArena = true;
      cg_point_vectors(s, A->vecArena, xexact_host != nullptr);
    } else {
      cg_point_vectors(s, (char*)sb_malloc(s->vecSlabBytes), xexact_host != nullptr);
    }
  }
  double* tmp = scratch_ws(0, m->nr);
  sb_h2d(tmp, b_host, nb);
  sb_permute(m, tmp, s->b);
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (xexact_host) {
    sb_h2d(tmp, xexact_host, nb);
    sb_permute(m, tmp, s->xexact);
    HIP_CHECK(hipStreamSynchronize(g.stream));
  }
  s->S         = (CgScalars*)sb_malloc(sizeof(CgScalars));
  HIP_CHECK(hipMemset(s->S, 0, sizeof(CgScalars))); // never an uninitialised failure flag (sb_cg_start points the push kernels at it)
  s->nPartials = (m->nr + 255) / 256;
  // level-0 partials: 4 per 256 rows; the tail beyond the last chunk stays +0.0
  s->partials = (double*)sb_malloc((4 * (size_t)s->nPartials + 4) * sizeof(double));
  HIP_CHECK(hipMemsetAsync(s->partials, 0, (4 * (size_t)s->nPartials + 4) * sizeof(double), g.stream));
  s->hist_cap  = 0;
  s->rr_hist = s->pAp_hist = nullptr;
  s->partials2 = (double*)sb_malloc((4 * (size_t)s->nPartials + 4) * sizeof(double));
  HIP_CHECK(hipMemsetAsync(s->partials2, 0, (4 * (size_t)s->nPartials + 4) * sizeof(double), g.stream));
#ifdef SB_LAB
  s->lead = (Lead*)sb_malloc(2 * sizeof(Lead));
  HIP_CHECK(hipMemsetAsync(s->lead, 0, 2 * sizeof(Lead), g.stream));
  s->vphase = (VPhase*)sb_malloc(sizeof(VPhase));
  HIP_CHECK(hipMemsetAsync(s->vphase, 0, sizeof(VPhase), g.stream));
#endif
  s->fused      = 1;
  s->use_graph  = 0;
  s->graphReady = false;
  s->iterGraph  = nullptr;
  s->timing     = false;
  s->evUsed     = 0;
  s->loop_ms    = 0.f;
  s->spmvTiming = false;
  s->spmvEvUsed = 0;
  s->k_next     = 1;
  s->started    = false;
  HIP_CHECK(hipEventCreate(&s->evLoop0));
  HIP_CHECK(hipEventCreate(&s->evLoop1));
  for (double& v : s->region_ms) v = 0.0;
  return s;
}

// (lab: tools/placement_lab2.py) device addresses of the loop's vectors: r, p (both buffers), Ap, x, b, partials
void sb_cg_debug_ptrs(const sb_cg* s, unsigned long long out[8])
{
  const void* v[8] = { s->r, s->pbuf[0], s->pbuf[1], s->Ap, s->x, s->b, s->partials, s->S };
  for (int i = 0; i < 8; i++) out[i] = (unsigned long long)v[i];
}

void sb_cg_free(sb_cg* s)
{
  if (!s) return;
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (s->halo && s->halo->p2p && s->halo->push.p2pErr == &s->S->p2p_error) {
    s->halo->push.p2pErr = nullptr;
    HIP_CHECK(hipMemcpy(s->halo->dPush, &s->halo->push, sizeof s->halo->push, hipMemcpyHostToDevice));
  }
  if (s->iterGraph) HIP_CHECK(hipGraphExecDestroy(s->iterGraph));
  for (hipEvent_t e : s->ev) HIP_CHECK(hipEventDestroy(e));
  for (hipEvent_t e : s->spmvEv) HIP_CHECK(hipEventDestroy(e));
  for (hipEvent_t e : s->phEv) HIP_CHECK(hipEventDestroy(e));
  HIP_CHECK(hipEventDestroy(s->evLoop0));
  HIP_CHECK(hipEventDestroy(s->evLoop1));
  if (s->vecFromArena) const_cast<sb_matrix*>(s->A)->vecArenaBusy = false; // r, Ap, x, b, both p buffers, xexact: back to the matrix
  else sb_free(s->vecSlab);
  sb_free(s->S), sb_free(s->partials), sb_free(s->rr_hist), sb_free(s->pAp_hist), sb_free(s->partials2), sb_free(s->vphase), sb_free(s->lead);
  delete s;
}

static void drop_graph(sb_cg* s)
{
  if (s->iterGraph) HIP_CHECK(hipGraphExecDestroy(s->iterGraph));
  s->iterGraph = nullptr, s->graphReady = false;
}

void sb_cg_set_fused(sb_cg* s, int fused)
{ // 0: the reference's op list; 1 (default): dots fused into their producers (5 launches per body).  Lab builds (-DSB_LAB)
  // additionally: 2: the vector phase of a body as one launch where that is possible (2 launches per body); 3: the two
  // scalar steps taken by workgroup 0 of their consumers (3 launches per body) -- both measured SLOWER at 128^3, see
  // below; the product treats every non-zero level as 1.
#ifndef SB_LAB
  fused = fused ? 1 : 0;
#endif
  if (s->fused != fused) drop_graph(s), s->vSP = -1, s->leadPlan = -1, s->fusepPlan = -1;
  s->fused = fused;
}

#ifdef SB_LAB

// Measured (MI355X, HPCG 128^3, Sell-64-256): 61.1 us per iteration against 51.5 us with the five launches.  The four
// kernels it replaces overlap their reads and writes freely (134 MB in ~20 us, Infinity-Cache assisted) and pay ~8 us
// for the two scalar launches; the one launch reads everything, THEN (after alpha) writes r and x, THEN (after beta)
// writes p: the two grid-wide waits serialise the traffic, and one 1024-thread workgroup per CU is a thin streaming
// configuration.  Kept selectable (sb_cg_set_fused(s, 2)) and tested, because the protocol -- workgroup 0 takes the
// scalar step while the others hold their elements in registers -- is what a persistent CG kernel would build on.
// The one-launch vector phase needs every workgroup of its grid resident at once (they wait for each other):
// the grid is what the occupancy calculation says fits, and the instantiation the smallest whose registers hold
// the rank's rows on that grid.  Not used when ranks share a GPU (SB_SHARED_GPU=1: rehearsal with several
// processes per device -- two such grids would wait for each other's CUs), when the all-reduce is not the
// in-kernel one, or with SB_VPHASE=0.
template <int SP, bool P2P> static bool vphase_try(sb_cg* s, uint32_t nSpans)
{
  int perCU = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, cg_vector_phase_k<SP, P2P>, 1024, 0));
  uint64_t grid = (uint64_t)perCU * (uint64_t)g.prop.multiProcessorCount;
  // (tests with several ranks on ONE device cap the grid so that all ranks' grids are resident together)
  if (const char* cap = getenv("SB_VPHASE_MAXGRID")) grid = std::min<uint64_t>(grid, (uint64_t)std::max(1, atoi(cap)));
  if (grid == 0 || (uint64_t)nSpans > 16ull * grid * SP) return false;
  s->vSP   = SP;
  s->vGrid = (uint32_t)std::min<uint64_t>(grid, (nSpans + 16ull * SP - 1) / (16ull * SP));
  return true;
}
static bool vphase_plan(sb_cg* s)
{
  if (s->vSP >= 0) return s->vSP > 0;
  s->vSP = 0;
  const bool off    = getenv("SB_VPHASE") && atoi(getenv("SB_VPHASE")) == 0;
  const bool shared = getenv("SB_SHARED_GPU") && atoi(getenv("SB_SHARED_GPU")) != 0;
  if (s->fused != 2 || off || s->nr == 0) return false;
  if (multi_rank() && (!p2p_dots() || shared)) return false;
  const uint32_t nSpans = ((s->nr + 255u) >> 8) * 2u;
  if (multi_rank()) return vphase_try<1, true>(s, nSpans) || vphase_try<2, true>(s, nSpans) || vphase_try<4, true>(s, nSpans);
  return vphase_try<1, false>(s, nSpans) || vphase_try<2, false>(s, nSpans) || vphase_try<4, false>(s, nSpans);
}
#else  // the product: five launches per body (or the reference's op list); DESIGN 4.4 has the measurements of the rest
static bool vphase_plan(sb_cg*) { return false; }
#endif // SB_LAB
int sb_cg_vector_phase(sb_cg* s) { return vphase_plan(s) ? s->vSP : 0; }

template <int MODE> static void scalar_launch(sb_cg* s, int defer_x, const double* q, int l1 = 0);
static int pAp_is_level1(const sb_cg* s);
static int fusealpha_plan(sb_cg* s, int l1, uint32_t vb);
static int fusebeta_plan(sb_cg* s, uint32_t vb);

#ifdef SB_LAB

// The scalar steps inside their consumers (sb_cg_set_fused(s, 3)): one rank only.
// Measured (MI355X, HPCG 128^3, Sell-64-256, same box, back to back): 58.6 us per iteration against 51.7 us with the
// five launches.  A dependent single-workgroup launch costs ~4 us here; waiting INSIDE a kernel for workgroup 0 --
// its partial loads, the reduction, an agent-scope store, the pollers' round trips, 500 workgroups resuming -- costs
// more (~7.5 us per step).  Not the default; kept selectable and tested (VERDICT r1 item 9 asked for 5 -> 3 launches).
static bool lead_plan(sb_cg* s)
{
  if (s->leadPlan >= 0) return s->leadPlan > 0;
  s->leadPlan = (s->fused == 3 && s->nr > 0 && !multi_rank()) ? 1 : 0;
  return s->leadPlan > 0;
}
#else
static bool lead_plan(sb_cg*) { return false; }
#endif // SB_LAB
static bool spmv_can_fuse_dot(const sb_cg* s);
// The p update inside the SpMV (pack.hip.h: spmv_prog_fusep; 4 launches per body: SpMV | alpha | r update | beta): where the
// matrix allows it (spmv_fusep_possible), in the default loop (fused = 1), on one rank or with the halo over peer-mapped
// memory (the push kernel then forms the boundary values itself; a send / recv exchange needs p in memory first).
// SB_FUSE_P=0 keeps the separate p update.
static bool fusep_plan(sb_cg* s)
{
  // Inside a solve the answer is the one sb_cg_start latched: the fused path keeps p double-buffered and picks the
  // buffer from the body count, the in-place path does not -- a body of the other kind in the middle of a solve (the matrix's
  // kernel mode or the wish changed between two sb_cg_run_iters pieces) would read the wrong p.  Such a change takes effect with
  // the next sb_cg_start.
  if (s->started && s->fusepLatched >= 0) return s->fusepLatched > 0;
  if (s->fusepPlan < 0) { // (the wish is decided once; whether it applies follows the matrix's kernel mode and the data plane)
    const char* env = getenv("SB_FUSE_P");
    s->fusepPlan    = (s->fusepWant >= 0 ? s->fusepWant != 0 : env ? atoi(env) != 0 : SB_FUSE_P_DEFAULT) ? 1 : 0;
  }
  bool ok = s->fusepPlan > 0 && s->fused == 1 && s->nr > 0 && !s->use_graph && spmv_fusep_possible(s->A);
  if (ok && multi_rank()) ok = s->halo ? halo_p2p_active(s->halo) : true;
  return ok;
}
// kernel launches per loop body.  One rank: 5 (p update | SpMV | alpha | r update | beta); 0 = the reference's op list.
// Several ranks add the halo kernels (peer-mapped: the push, 0 with the push inside the SpMV launch, + a pull where the
// SpMV is not the pattern kernel; otherwise the pack kernel in front of the send / recv group) and, without the in-kernel
// all-reduce, one more kernel per dot (local reduce | all-reduce | scalar step) -- the all-reduce / send-recv calls
// themselves are counted by sb_cg_collectives_per_body.
int sb_cg_launches_per_body(sb_cg* s)
{
  int base = vphase_plan(s) ? 2 : lead_plan(s) ? 3 : fusep_plan(s) ? 4 : s->fused ? 5 : 0;
  if (base >= 4 && fusealpha_plan(s, fusep_plan(s) ? 1 : pAp_is_level1(s), 1024u)) base -= 1; // (alpha step inside the r update)
  if (base >= 4 && fusebeta_plan(s, 1024u)) base -= 1;                                          // (beta step inside the p update)
  if (!multi_rank() || base == 0) return base;
  int n = base;
  if (s->halo) {
    const bool inSpmv = halo_p2p_active(s->halo) && spmv_can_fuse_dot(s) && spmv_uses_patterns(s->A);
    if (inSpmv) n += (s->halo->totalSend && !g.pushInside) ? 1 : 0;
    else if (halo_p2p_active(s->halo)) n += (s->halo->totalSend ? 1 : 0) + (s->halo->indegree ? 1 : 0);
    else n += s->halo->totalSend ? 1 : 0;
  }
  if (!p2p_dots()) n += 2;
  return n;
}
// communicator calls per loop body (RCCL / transport): 2 all-reduces + 1 send-recv group without the peer-mapped paths
int sb_cg_collectives_per_body(sb_cg* s)
{
  if (!multi_rank() || !s->fused) return 0;
  return (p2p_dots() ? 0 : 2) + (s->halo && !halo_p2p_active(s->halo) ? 1 : 0);
}

#ifdef SB_LAB

static long long lead_timeout() { return 2000ll * P2P_TICKS_PER_MS; }
template <typename K> static dim3 lead_grid(K kernel, uint32_t work, uint32_t perBlock)
{ // one round of resident 1024-thread workgroups (what the occupancy calculation says fits: 1 per CU at ~96 VGPRs)
  int perCU = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, 1024, 0));
  const uint32_t cap = (uint32_t)g.prop.multiProcessorCount * (uint32_t)std::max(1, perCU);
  return dim3(std::max(1u, std::min(cap, (work + perBlock - 1) / perBlock)));
}
// alpha step + r update + r.r partials (-> partials2)
static void launch_lead_r(sb_cg* s)
{
  const uint32_t nSpans = ((s->nr + 255u) >> 8) * 2u;
  hipLaunchKernelGGL(cg_lead_r_k, lead_grid(cg_lead_r_k, nSpans, 32) /* two spans per wave and step */, dim3(1024), 0, g.stream, s->nr, s->Ap,
      s->r, s->S, s->partials, s->partials2, s->nPartials, s->rr_hist, s->pAp_hist, s->lead + 0, lead_timeout(), pAp_is_level1(s));
  HIP_CHECK(hipGetLastError());
  s->betaOwed = true;
}
// beta step / loop test + p update + owed x update
static void launch_lead_p(sb_cg* s)
{
  hipLaunchKernelGGL(cg_lead_p_k, lead_grid(cg_lead_p_k, s->nr / 2 + 1, 2048) /* two element pairs per thread and step */, dim3(1024), 0,
      g.stream, s->nr, s->r, s->p, s->x, s->S, s->partials2, s->nPartials, s->rr_hist, s->pAp_hist, s->lead + 1, lead_timeout());
  HIP_CHECK(hipGetLastError());
  s->betaOwed = false;
}
// the beta step / loop test of the last enqueued body as its own launch (nobody's p update follows yet)
static void flush_beta(sb_cg* s)
{
  if (!s->betaOwed) return;
  scalar_launch<1>(s, 1, s->partials2);
  phase_mark(s, PH_BETA);
  s->betaOwed = false;
}

static void launch_vphase(sb_cg* s)
{
  const long long ticks = 2000ll * P2P_TICKS_PER_MS + (multi_rank() ? g.p2pTimeoutTicks : 0ll);
  unsigned long long seq = 0;
  if (multi_rank()) seq = g.p2pSeq + 1ull, g.p2pSeq += 2ull; // two all-reduces per launch
#define VP_LAUNCH(SPN, PP)                                                                                              \
  hipLaunchKernelGGL((cg_vector_phase_k<SPN, PP>), dim3(s->vGrid), dim3(1024), 0, g.stream, s->nr, s->r, s->p, s->Ap, s->x, \
      s->S, s->partials, s->partials2, s->nPartials, s->rr_hist, s->pAp_hist, s->vphase, ticks, (const P2PView*)g.p2pView, seq, pAp_is_level1(s))
  if (multi_rank()) {
    if (s->vSP == 1) VP_LAUNCH(1, true);
    else if (s->vSP == 2) VP_LAUNCH(2, true);
    else VP_LAUNCH(4, true);
  } else {
    if (s->vSP == 1) VP_LAUNCH(1, false);
    else if (s->vSP == 2) VP_LAUNCH(2, false);
    else VP_LAUNCH(4, false);
  }
#undef VP_LAUNCH
  HIP_CHECK(hipGetLastError());
}
#else
static void launch_lead_r(sb_cg*) {}
static void launch_lead_p(sb_cg*) {}
static void flush_beta(sb_cg*) {}
static void launch_vphase(sb_cg*) {}
#endif // SB_LAB
// hipGraph replay of a loop body was measured slower (-7 % at 128^3, -13 ... -35 % at 64^3: DESIGN 4.4): lab builds only
// 1 / 0: take / do not take the p update inside the SpMV where the matrix and the data plane allow it; -1: the default
// (SB_FUSE_P, else the library's).  sb_cg_fuse_p: what the loop will do.
void sb_cg_set_fuse_p(sb_cg* s, int on)
{
  s->fusepWant = on < 0 ? -1 : on != 0;
  s->fusepPlan = -1;
}
int sb_cg_fuse_p(sb_cg* s) { return fusep_plan(s) ? 1 : 0; }

void sb_cg_set_graph(sb_cg* s, int use_graph)
{
#ifdef SB_LAB
  s->use_graph = use_graph;
#else
  (void)use_graph;
  s->use_graph = 0;
#endif
}

void sb_cg_spmv_timing(sb_cg* s, int on)
{
  s->spmvTiming = on != 0;
  s->spmvEvUsed = 0;
}

double sb_cg_spmv_ms(sb_cg* s, int* launches)
{ // sum of the event-bracketed SpMV launches since sb_cg_spmv_timing(s, 1)
  HIP_CHECK(hipStreamSynchronize(g.stream));
  double total = 0.0;
  int n        = 0;
  for (size_t i = 0; i + 1 < s->spmvEvUsed; i += 2) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, s->spmvEv[i], s->spmvEv[i + 1]));
    total += ms;
    n++;
  }
  if (launches) *launches = n;
  return total;
}

// lab call (tools/placement_lab11.py): the event-bracketed SpMV launches since sb_cg_spmv_timing(s, 1), one by one (us); returns
// how many there are (at most `cap` are written)
int sb_cg_spmv_us_series(sb_cg* s, float* out, int cap)
{
  HIP_CHECK(hipStreamSynchronize(g.stream));
  int n = 0;
  for (size_t i = 0; i + 1 < s->spmvEvUsed; i += 2, n++) {
    if (n >= cap) continue;
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, s->spmvEv[i], s->spmvEv[i + 1]));
    out[n] = 1e3f * ms;
  }
  return n;
}

void sb_cg_phase_timing(sb_cg* s, int on)
{
  s->phaseTiming = on != 0;
  s->phUsed      = 0;
}

int sb_cg_phase_ms(sb_cg* s, double ms_out[8], int count_out[8])
{ // summed duration and number of occurrences of every phase since sb_cg_phase_timing(s, 1); returns the number of phases
  HIP_CHECK(hipStreamSynchronize(g.stream));
  for (int i = 0; i < 8; i++) ms_out[i] = 0.0, count_out[i] = 0;
  for (size_t i = 1; i < s->phUsed; i++) {
    const int ph = s->phId[i];
    if (ph < 0) continue; // a start marker: the time in front of it (host gaps between run_iters calls) belongs to nobody
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, s->phEv[i - 1], s->phEv[i]));
    ms_out[ph] += ms, count_out[ph]++;
  }
  return PH_COUNT;
}

void sb_cg_counters(const sb_cg* s, int out[5])
{ // stop, stop_next, iters, n_rr, n_pAp of the device control block
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h.stop, out[1] = h.stop_next, out[2] = h.iters, out[3] = h.n_rr, out[4] = h.n_pAp;
}

static bool spmv_can_fuse_dot(const sb_cg* s)
{ // p.Ap partials in the SpMV epilogue: the wave-per-chunk kernels (SCS C=64, or CRS through its mirror)
  return s->fused && (s->A->fmt == 1 ? s->A->C == 64 : spmv_uses_patterns(s->A));
}

// levels 1-2 of the reduction + the scalar step: one 1-workgroup launch after the producer
// (several ranks: local sum -> RCCL all-reduce in place on the stream -> scalar step;
// MPI_Allreduce of src/comm.c:659)
// l1: q holds level-1 values (cg_update_r_k) instead of level-0 partials
template <int MODE> static void scalar_launch(sb_cg* s, int defer_x, const double* q, int l1)
{
  if (!q) q = s->partials;
  if (multi_rank() && p2p_dots()) { // local reduce, in-kernel all-reduce and scalar step in ONE launch
    hipLaunchKernelGGL((cg_scalar_p2p_k<MODE>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, q,
        s->S, s->rr_hist, s->pAp_hist, defer_x, (const P2PView*)g.p2pView, ++g.p2pSeq, l1,
        (const int*)(s->halo && s->halo->p2p ? s->halo->err : nullptr));
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL((cg_scalar_k<MODE, true>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, q,
      s->S, s->rr_hist, s->pAp_hist, multi_rank() ? 1 : 0, defer_x, l1);
  HIP_CHECK(hipGetLastError());
  if (multi_rank()) {
    mark(s, R_DDOT);
    sb_comm_reduction(&s->S->local, 1);
    mark(s, R_COMM);
    hipLaunchKernelGGL((cg_scalar_k<MODE, false>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, q,
        s->S, s->rr_hist, s->pAp_hist, 0, defer_x, 0);
    HIP_CHECK(hipGetLastError());
  }
}

// 1: the SpMV's fused dot wrote LEVEL-1 values of p.Ap (one per 256 rows), 0: level-0 partials (dot pass, lab kernels)
// (fused loop behind a kernel without a dot of its own: the dot pass is dot_l1_k)
static int pAp_is_level1(const sb_cg* s) { return spmv_can_fuse_dot(s) ? (spmv_dot_kind(s->A) == 2 ? 1 : 0) : (s->fused && s->nr ? 1 : 0); }

// The scalar steps inside their consumers' launches (kernels.hip.h: cg_update_r_k<ALPHA>, cg_update_p<BETA>): EVERY workgroup of
// the consumer takes the step itself -- nobody waits for anybody --, workgroup 0 records it.  Returns the mode: 0 the separate
// scalar launch; 1 (one rank) each workgroup reduces the producer's level-1 values itself; 2 (several ranks on the
// communicator's collectives) the sum is already reduced and all-reduced into S->local by the two launches before, the third
// launch of that dot goes.  The peer-mapped plane keeps its one-launch step (local reduce, exchange and step: cg_scalar_p2p_k).
// Needs level-1 values from the producer and 1024-thread workgroups.  SB_FUSE_ALPHA=0 / sb_cg_set_fuse_alpha(s, 0) and
// SB_FUSE_BETA=0 / sb_cg_set_fuse_beta(s, 0): the separate launches.
static int fold_mode(sb_cg* s, int l1, uint32_t vb)
{
  if (!s->fused || !l1 || vb != 1024u || vphase_plan(s) || lead_plan(s)) return 0;
  if (multi_rank()) return p2p_dots() ? 0 : 2;
  return 1;
}
static int fusealpha_plan(sb_cg* s, int l1, uint32_t vb)
{
  static const int env = getenv("SB_FUSE_ALPHA") ? atoi(getenv("SB_FUSE_ALPHA")) != 0 : -1;
  const bool want = s->fuseAlphaWant >= 0 ? s->fuseAlphaWant != 0 : env >= 0 ? env != 0 : true;
  return want ? fold_mode(s, l1, vb) : 0;
}
// (the beta step rides in the p update's launch: only where the p update is a launch of its own, i.e. not inside the SpMV)
static int fusebeta_plan(sb_cg* s, uint32_t vb)
{
  static const int env = getenv("SB_FUSE_BETA") ? atoi(getenv("SB_FUSE_BETA")) != 0 : -1;
  const bool want = s->fuseBetaWant >= 0 ? s->fuseBetaWant != 0 : env >= 0 ? env != 0 : true;
  if (!want || fusep_plan(s) || s->use_graph) return 0;
  return fold_mode(s, 1, vb);
}
void sb_cg_set_fuse_alpha(sb_cg* s, int on) { s->fuseAlphaWant = on < 0 ? -1 : on != 0; }
void sb_cg_set_fuse_beta(sb_cg* s, int on) { s->fuseBetaWant = on < 0 ? -1 : on != 0; }

// several ranks, communicator's collectives: local levels 1-2 of a dot into S->local, all-reduced in place on the stream
// (MPI_Allreduce of src/comm.c:659); the step itself is taken by the consumer (fold mode 2) or by cg_scalar_k<MODE, false>
template <int MODE> static void scalar_reduce_only(sb_cg* s, const double* q, int l1)
{
  hipLaunchKernelGGL((cg_scalar_k<MODE, true>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, q, s->S, s->rr_hist, s->pAp_hist, 1, 0, l1);
  HIP_CHECK(hipGetLastError());
  mark(s, R_DDOT);
  sb_comm_reduction(&s->S->local, 1);
  mark(s, R_COMM);
}

// alpha step (src/CGSolver.c:124-126) and r -= alpha Ap + the r.r values of the next body (:128, :112): two launches, or one
static void alpha_and_r_update(sb_cg* s, int l1, uint32_t capV, uint32_t vb, const int* stop)
{
  const uint32_t n = s->nr;
  const dim3 grid(std::max(1u, std::min(capV, (((n + 255u) >> 8) + vb / 64 - 1) / (vb / 64))));
  const int mode = fusealpha_plan(s, l1, vb);
  if (mode) {
    // one workgroup per CU (the separate r update runs two): every workgroup reads all m values, so half the workgroups is half
    // that traffic -- 128^3, same box, alternating: 41.6 / 42.1 us per iteration against 42.4 / 42.7 with two (and 43.8 / 44.2
    // with the separate alpha launch)
    static const uint32_t perCu = getenv("SB_ALPHA_WG_PER_CU") ? (uint32_t)std::max(1, atoi(getenv("SB_ALPHA_WG_PER_CU"))) : 1u;
    const dim3 gridA(std::max(1u, std::min((uint32_t)g.prop.multiProcessorCount * perCu, grid.x)));
    if (mode == 2) {
      scalar_reduce_only<2>(s, s->partials, l1);
      phase_mark(s, PH_ALPHA);
      hipLaunchKernelGGL(cg_update_r_k<2>, grid, dim3(vb), 0, g.stream, n, s->Ap, s->r, s->S, s->partials2, stop, s->nPartials,
          (const double*)s->partials, s->rr_hist, s->pAp_hist);
    } else
      hipLaunchKernelGGL(cg_update_r_k<1>, gridA, dim3(vb), 0, g.stream, n, s->Ap, s->r, s->S, s->partials2, stop, s->nPartials,
          (const double*)s->partials, s->rr_hist, s->pAp_hist);
    HIP_CHECK(hipGetLastError());
    mark(s, R_WAXPBY);
    phase_mark(s, PH_R_UPDATE);
    return;
  }
  scalar_launch<2>(s, 0, nullptr, l1);
  mark(s, R_DDOT);
  phase_mark(s, PH_ALPHA);
  hipLaunchKernelGGL(cg_update_r_k<0>, grid, dim3(vb), 0, g.stream, n, s->Ap, s->r, s->S, s->partials2, stop, 0u,
      (const double*)nullptr, (double*)nullptr, (double*)nullptr);
  HIP_CHECK(hipGetLastError());
  mark(s, R_WAXPBY);
  phase_mark(s, PH_R_UPDATE);
}

// beta step / loop test behind the r update (:107, :111-113, :116): its own launch(es), or left owing to the next body's p
// update (cg_update_p<BETA>; flush_beta_fold takes it where no body follows)
static void beta_step_or_owe(sb_cg* s, uint32_t vb)
{
  const int mode = fusebeta_plan(s, vb);
  if (mode) {
    if (mode == 2) scalar_reduce_only<1>(s, s->partials2, 1);
    s->betaFold = mode;
    return;
  }
  scalar_launch<1>(s, 1, s->partials2, 1);
  mark(s, R_DDOT);
  phase_mark(s, PH_BETA);
}
static void flush_beta_fold(sb_cg* s)
{
  if (!s->betaFold) return;
  if (s->betaFold == 2) { // r.r is in S->local already
    hipLaunchKernelGGL((cg_scalar_k<1, false>), dim3(1), dim3(1024), 0, g.stream, s->nPartials, (const double*)s->partials2, s->S,
        s->rr_hist, s->pAp_hist, 0, 1, 0);
    HIP_CHECK(hipGetLastError());
  } else scalar_launch<1>(s, 1, s->partials2, 1);
  mark(s, R_DDOT);
  phase_mark(s, PH_BETA);
  s->betaFold = 0;
}

static void spmv_event(sb_cg* s)
{
  if (!s->spmvTiming) return;
  if (s->spmvEvUsed == s->spmvEv.size()) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, timing_event_flags()));
    s->spmvEv.push_back(e);
  }
  HIP_CHECK(hipEventRecord(s->spmvEv[s->spmvEvUsed++], g.stream));
}

// The SpMV launch that follows is the one to time: bracket it with two events recorded on the stream.  The events are created
// with hipEventDisableSystemFence (a default event performs a system-scope fence -- L2 write-back and invalidate -- when it is
// recorded, which costs time of its own and slows the work behind it; the HIP headers recommend the flag for timing).
// Measured on one box, same run (tools/ext_timing_check.sh; rocprofv3 of the same loop: 129.8 / 30.7 us on another box):
//   reference-layout SpMV / fused SpMV:  record, no fence 135.7 / 29.3 us;  record, default events 137.7 / 31.3 us;
//   hipExtLaunchKernelGGL's start / stop stamps of the launch itself (SB_SPMV_TIMING=ext) 138.9 / 27.6 and 139.5 / 30.2 us.
// None of them reaches rocprofv3's own begin -> end of the dispatch for the long kernel (it reports 129.8 us for the launches
// of the clean passes, the events pass and the phases pass alike: the kernel is not disturbed, the brackets are wider).
static bool spmv_time_begin(sb_cg* s)
{
  if (!s->spmvTiming) return false;
  static const bool record = !(getenv("SB_SPMV_TIMING") && strcmp(getenv("SB_SPMV_TIMING"), "ext") == 0);
  if (record) {
    spmv_event(s);
    return true;
  }
  while (s->spmvEv.size() < s->spmvEvUsed + 2) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, timing_event_flags()));
    s->spmvEv.push_back(e);
  }
  g_spmvEvA = s->spmvEv[s->spmvEvUsed], g_spmvEvB = s->spmvEv[s->spmvEvUsed + 1];
  s->spmvEvUsed += 2;
  return true;
}
static void spmv_time_end(sb_cg* s)
{
  if (!s->spmvTiming) return;
  if (g_spmvEvA) g_spmvEvA = g_spmvEvB = nullptr;
  else spmv_event(s);
}

// Ap = A p and the level-0 partials of p.Ap (src/CGSolver.c:123-125): fused into the SpMV epilogue
// where the kernel is wave-per-chunk, otherwise SpMV then a dot pass
static void spmv_and_pAp(sb_cg* s, const int* stop)
{
  const uint32_t n = s->nr;
  if (spmv_can_fuse_dot(s)) {
    spmv_time_begin(s);
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop);
    spmv_time_end(s);
    mark(s, R_SPMVM);
    phase_mark(s, PH_SPMV);
  } else {
    spmv_time_begin(s);
    launch_spmv(s->A, s->p, s->Ap, nullptr, stop);
    spmv_time_end(s);
    mark(s, R_SPMVM);
    phase_mark(s, PH_SPMV);
    if (s->fused && n) { // level-1 values straight away (pAp_is_level1): a quarter of the bytes for the step, which can then ride in the r update
      const uint32_t nGroups = (n + 255u) >> 8;
      hipLaunchKernelGGL(dot_l1_k, dim3(std::max(1u, std::min((uint32_t)g.prop.multiProcessorCount * 2u, (nGroups + 15u) / 16u))), dim3(1024), 0,
          g.stream, n, (const double*)s->p, (const double*)s->Ap, s->partials, stop);
      HIP_CHECK(hipGetLastError());
    } else launch_dot_spans(0, n, s->p, s->Ap, nullptr, nullptr, s->S, s->partials, stop);
    phase_mark(s, PH_DOT_PASS);
  }
}

// one loop body of solveCG (src/CGSolver.c:108-128).  Fused path: the r.r partials of the
// NEXT body come out of this body's x/r update, and its beta + loop test are taken right
// after it, so a body is: p update | SpMV (+p.Ap partials) | alpha | x/r update (+r.r
// partials) | beta, loop test.
static void loop_body(sb_cg* s, int k)
{
  const uint32_t n = s->nr;
  const int* stop  = &s->S->stop;
  // workgroups of the two vector kernels: 1024 threads (512 workgroups to dispatch instead of 2048: +0.7 % it/s at 128^3,
  // measured back to back; SB_VEC_BLOCK=256|512 for comparison)
  static const uint32_t vb = getenv("SB_VEC_BLOCK") && atoi(getenv("SB_VEC_BLOCK")) >= 64 ? (uint32_t)atoi(getenv("SB_VEC_BLOCK")) & ~63u : 1024u;
  const uint32_t capV = (uint32_t)g.prop.multiProcessorCount * (2048u / vb);
  dim3 gridV(std::max(1u, std::min(capV, (n / 2 + 1 + vb - 1) / vb))), blockV(vb);
  if (fusep_plan(s)) {
    // p = r + beta p (:114; k = 1: p = r, :109), the owed x update (:127), the halo exchange (:122) and Ap = A p with its p.Ap
    // values (:123-125) in ONE launch (+ the halo push on several ranks): body k reads p_{k-1} in pbuf[(k - 1) & 1] and writes
    // p_k into pbuf[k & 1]
    const int which    = k == 1;
    const double* pold = which ? s->r : s->pbuf[(k - 1) & 1];
    double* pnew       = s->pbuf[k & 1];
    HaloWait hw;
    memset(&hw, 0, sizeof hw);
    const bool halo = multi_rank() && s->halo;
    if (halo) {
      sb_halo* h = s->halo;
      const unsigned long long seq = ++h->seq;
      if (g.pushInside) hw.push = h->dPush, hw.nPush = h->totalSend ? 16u : 0u;
      else if (h->totalSend) {
        hipLaunchKernelGGL(halo_push_fusep_k, dim3(stream_grid(h->totalSend, 256)), dim3(256), 0, g.stream, h->push, pold, s->r,
            (const CgScalars*)s->S, which, seq);
        HIP_CHECK(hipGetLastError());
      }
      phase_mark(s, PH_HALO);
      hw.flags = h->stage + 2 * (size_t)h->externalCount;
      hw.ext   = reinterpret_cast<const double*>(h->stage + (seq & 1ull) * (size_t)h->externalCount);
      hw.src = h->dSrcRank, hw.nsrc = h->indegree, hw.seq = seq, hw.err = h->err;
      hw.stopw = &s->S->stop, hw.timeoutTicks = h->push.timeoutTicks;
    }
    spmv_time_begin(s);
    launch_spmv_fusep(s->A, pold, s->r, pnew, s->x, s->Ap, s->S, which, s->partials, halo ? &hw : nullptr);
    spmv_time_end(s);
    phase_mark(s, PH_SPMV);
    s->p = pnew; // (host-side view: the newest p; cg_x_finalize picks the buffer of the last body that RAN from the device counters)
    alpha_and_r_update(s, 1, capV, vb, stop);
    scalar_launch<1>(s, 1, s->partials2, 1);
    phase_mark(s, PH_BETA);
    return;
  }
  if (k == 1) {
    if (n) hipLaunchKernelGGL(cg_update_p<0>, gridV, blockV, 0, g.stream, n, s->r, s->p, (double*)nullptr, s->S, 1, 0u, (const double*)nullptr, (double*)nullptr); // p = r (:109)
    mark(s, R_WAXPBY);
    phase_mark(s, PH_P_UPDATE);
  } else if (vphase_plan(s)) {
    // p = r + beta p (:114) was taken at the end of the previous body's vector phase
  } else if (s->betaOwed) { // (lead kernels) the previous body's beta step / loop test rides in front of the p update
    launch_lead_p(s);
    phase_mark(s, PH_P_UPDATE);
  } else {
    if (!s->fused) { // rtrans = r.r ; beta (:111-113)
      launch_dot_spans(0, n, s->r, s->r, nullptr, nullptr, s->S, s->partials, stop);
      phase_mark(s, PH_DOT_PASS);
      scalar_launch<1>(s, 0, nullptr);
      mark(s, R_DDOT);
      phase_mark(s, PH_BETA);
    }
    // p = r + beta p (:114); fused path: also the x update owed by the previous body (:127) -- and, where the previous body left
    // its beta step / loop test owing, that step at the head of this launch
    if (s->betaFold == 2)
      hipLaunchKernelGGL(cg_update_p<2>, gridV, blockV, 0, g.stream, n, s->r, s->p, s->x, s->S, 0, s->nPartials, (const double*)s->partials2, s->rr_hist);
    else if (s->betaFold == 1)
      hipLaunchKernelGGL(cg_update_p<1>, gridV, blockV, 0, g.stream, n, s->r, s->p, s->x, s->S, 0, s->nPartials, (const double*)s->partials2, s->rr_hist);
    else if (n)
      hipLaunchKernelGGL(cg_update_p<0>, gridV, blockV, 0, g.stream, n, s->r, s->p, s->fused ? s->x : (double*)nullptr, s->S, 0, 0u,
          (const double*)nullptr, (double*)nullptr);
    s->betaFold = 0;
    mark(s, R_WAXPBY);
    phase_mark(s, PH_P_UPDATE);
  }
  HIP_CHECK(hipGetLastError());
  // Off by default: a cross-stream event dependency costs ~12 us on this platform (measured with
  // an x update moved beside the beta step: 63 -> 88 us per iteration for one fork + join), which
  // is about what the overlap can hide.  SB_HALO_OVERLAP=1 enables it.
#ifdef SB_LAB
  static const bool overlapHalo = getenv("SB_HALO_OVERLAP") && atoi(getenv("SB_HALO_OVERLAP")) != 0;
#else
  const bool overlapHalo = false; // (measured: one cross-stream dependency costs what the overlap hides; lab builds only)
#endif
  if (overlapHalo && multi_rank() && s->halo && spmv_can_fuse_dot(s) && spmv_can_split(s->A)) {
    // :122-126 with the halo exchange hidden behind the interior tiles: the exchange (pack,
    // send/recv into the tail of p) AND the few halo-touching tiles that need it run on a
    // second stream while the tiles that touch no halo column are multiplied on the main one;
    // the scalar step waits for both.  RCCL calls on the one communicator stay ordered: the
    // exchange is complete (event) before anything later on the main stream.
    mark(s, R_COMM);
    spmv_event(s);
    // (a host-mediated transport blocks the host inside halo_exchange, so nothing overlaps
    //  there, but the fork / join is the same code)
    HIP_CHECK(hipEventRecord(g.evFork, g.stream));
    HIP_CHECK(hipStreamWaitEvent(g.stream2, g.evFork, 0));
    halo_exchange(s->halo, s->p, stop, g.stream2, true);
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop, 2, g.stream2);
    HIP_CHECK(hipEventRecord(g.evJoin, g.stream2));
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop, 1);
    HIP_CHECK(hipStreamWaitEvent(g.stream, g.evJoin, 0));
    spmv_event(s);
    mark(s, R_SPMVM);
    phase_mark(s, PH_SPMV);
  } else if (multi_rank() && halo_p2p_active(s->halo) && spmv_can_fuse_dot(s) && spmv_uses_patterns(s->A)) {
    // :122-126 over peer-mapped memory with the pull inside the SpMV: the halo-touching tiles (stored
    // last) wait for the neighbours' pushes themselves and read the staging area; interior tiles hide it
    // SB_HALO_PUSH_INSIDE=1: the push rides in the SpMV launch too (its first 16 workgroups) instead of a launch of its
    // own.  Bit-identical (tests/test_gpu_multirank.py runs both); NOT the default: the only place it can be timed here
    // is N ranks sharing one GPU, where it is slower (2 x 128^3: 291 vs 178 us per step -- a rank's waiting tiles keep
    // the other rank's SpMV, and with it its push, off the CUs), and on ranks with a GPU each, where it should save the
    // ~4.5 us launch, it cannot be measured from this box.
    sb_halo* h = s->halo;
    const bool pushInside = g.pushInside;
    HaloWait hw;
    memset(&hw, 0, sizeof hw);
    if (pushInside) {
      ++h->seq;
      hw.push = h->dPush, hw.nPush = h->totalSend ? 16u : 0u;
    } else halo_exchange(h, s->p, stop, nullptr, true, true);
    mark(s, R_COMM);
    phase_mark(s, PH_HALO);
    hw.flags = h->stage + 2 * (size_t)h->externalCount;
    hw.ext   = reinterpret_cast<const double*>(h->stage + (h->seq & 1ull) * (size_t)h->externalCount);
    hw.src = h->dSrcRank, hw.nsrc = h->indegree, hw.seq = h->seq, hw.err = h->err;
    hw.stopw = &s->S->stop, hw.timeoutTicks = h->push.timeoutTicks;
    spmv_time_begin(s);
    launch_spmv(s->A, s->p, s->Ap, s->partials, stop, 0, nullptr, &hw);
    spmv_time_end(s);
    mark(s, R_SPMVM);
    phase_mark(s, PH_SPMV);
  } else {
    halo_exchange(s->halo, s->p, stop, nullptr, true); // :122
    mark(s, R_COMM);
    if (multi_rank() && s->halo) phase_mark(s, PH_HALO);
    spmv_and_pAp(s, stop);
  }
  if (vphase_plan(s)) { // alpha | x, r update + r.r | beta, loop test | the next body's p update: one launch
    launch_vphase(s);
    phase_mark(s, PH_R_UPDATE);
    return;
  }
  if (lead_plan(s)) { // alpha step in front of the r update; the beta step waits for the next body's p update (or flush_beta)
    launch_lead_r(s);
    phase_mark(s, PH_R_UPDATE);
    return;
  }
  if (s->fused) { // alpha; r -= alpha Ap (:128) + next r.r, beta, loop test; x += alpha p (:127) is owed
    // (level-1 values of r.r into partials2: `partials` keeps the layout the p.Ap producers write)
    alpha_and_r_update(s, pAp_is_level1(s), capV, vb, stop);
    beta_step_or_owe(s, vb);
    return;
  }
  scalar_launch<2>(s, 0, nullptr, pAp_is_level1(s));
  mark(s, R_DDOT);
  phase_mark(s, PH_ALPHA);
  if (n) {
    const dim3 gridW(stream_grid(n / 2 + 1, 256)), blockW(256); // (the reference-shaped ops keep their 256-thread workgroups)
    hipLaunchKernelGGL(waxpby_sdev_k, gridW, blockW, 0, g.stream, n, s->x, &s->S->alpha, s->p, s->x, stop);
    hipLaunchKernelGGL(waxpby_sdev_k, gridW, blockW, 0, g.stream, n, s->r, &s->S->neg_alpha, s->Ap, s->r, stop);
    HIP_CHECK(hipGetLastError());
    mark(s, R_WAXPBY);
    phase_mark(s, PH_R_UPDATE);
  }
}

static void ensure_hist(sb_cg* s, int cap)
{
  if (cap <= s->hist_cap) return;
  sb_free(s->rr_hist), sb_free(s->pAp_hist);
  s->hist_cap = cap;
  s->rr_hist  = (double*)sb_malloc((size_t)cap * sizeof(double));
  s->pAp_hist = (double*)sb_malloc((size_t)cap * sizeof(double));
  drop_graph(s); // captured pointers are stale
}

static void run_body_maybe_graph(sb_cg* s, int k)
{ // k >= 2 bodies are iteration-invariant (k lives in the device control block)
  // (lead kernels: the captured body is the chained one, whose p update carries the previous body's beta step)
  if (k < 2 || !s->use_graph || multi_rank() || s->timing || s->spmvTiming || s->phaseTiming || (lead_plan(s) && !vphase_plan(s) && !s->betaOwed)) {
    loop_body(s, k);
    return;
  }
  if (!s->graphReady) {
    hipGraph_t graph;
    HIP_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
    loop_body(s, 2);
    HIP_CHECK(hipStreamEndCapture(g.stream, &graph));
    // (captured with betaOwed set, and the body leaves it set: nothing to restore)
    HIP_CHECK(hipGraphInstantiate(&s->iterGraph, graph, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(graph));
    s->graphReady = true;
  }
  HIP_CHECK(hipGraphLaunch(s->iterGraph, g.stream));
}

void sb_cg_start(sb_cg* s, int itermax, double eps)
{
  need_init();
  const uint32_t n = s->nr;
  s->started      = false;
  s->fusepLatched = fusep_plan(s) ? 1 : 0; // decided once per solve (fusep_plan)
  if (s->halo && s->halo->p2p && s->halo->push.p2pErr != &s->S->p2p_error) {
    // the push kernels of THIS solve also look at its control block's failure flag (poisoning: kernels.hip.h).  Set per solve,
    // not at create: the plan is shared, and a second sb_cg on the same halo must not redirect a running loop's pushes to a
    // control block that is not in use
    s->halo->push.p2pErr = &s->S->p2p_error;
    HIP_CHECK(hipStreamSynchronize(g.stream));
    HIP_CHECK(hipMemcpy(s->halo->dPush, &s->halo->push, sizeof s->halo->push, hipMemcpyHostToDevice));
  }
  ensure_hist(s, itermax + 2);
  s->timing  = !s->fused; // the reference-shaped op list is the one that gets the region table
  s->evUsed  = 0;
  memset(&s->hostS, 0, sizeof s->hostS);
  s->hostS.itermax  = itermax;
  s->hostS.eps      = eps;
  s->hostS.hist_cap = s->hist_cap;
  HIP_CHECK(hipMemcpyAsync(s->S, &s->hostS, sizeof(CgScalars), hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipMemsetAsync(s->x, 0, (size_t)n * sizeof(double), g.stream)); // x0 = 0 (:28)
  s->p = s->pbuf[0];
  HIP_CHECK(hipMemsetAsync(s->pbuf[0], 0, (size_t)s->nc * sizeof(double), g.stream));
  HIP_CHECK(hipMemsetAsync(s->pbuf[1], 0, (size_t)s->nc * sizeof(double), g.stream));
  mark(s, -1);
  // prologue, src/CGSolver.c:94-100
  launch_waxpby(n, 1.0, s->x, 0.0, s->x, s->p, nullptr);
  mark(s, R_WAXPBY);
  halo_exchange(s->halo, s->p, nullptr);
  mark(s, R_COMM);
  launch_spmv(s->A, s->p, s->Ap, nullptr, nullptr);
  mark(s, R_SPMVM);
  if (s->fused) {
    launch_dot_spans(2, n, s->b, s->Ap, nullptr, s->r, s->S, s->partials, nullptr);
    mark(s, R_WAXPBY);
  } else {
    launch_waxpby(n, 1.0, s->b, -1.0, s->Ap, s->r, nullptr);
    mark(s, R_WAXPBY);
    launch_dot_spans(0, n, s->r, s->r, nullptr, nullptr, s->S, s->partials, nullptr);
  }
  scalar_launch<0>(s, 0, nullptr);
  mark(s, R_DDOT);
  s->k_next   = 1;
  s->started  = true;
  s->betaOwed = false;
  s->betaFold = 0;
}

void sb_cg_run_iters(sb_cg* s, int iters)
{
  need_init();
  if (!s->started) SB_FATAL("sb_cg_run_iters before sb_cg_start");
  phase_mark(s, -1);
  for (int i = 0; i < iters; i++) run_body_maybe_graph(s, s->k_next++);
  flush_beta(s); // every call leaves the loop state complete (counters, history, stop flag)
  flush_beta_fold(s);
}

int sb_cg_finish(sb_cg* s)
{
  need_init();
  if (s->nr) { // the x update the last body left to "the next p update": nobody comes after it
    // (fused p update: body k left p_k in pbuf[k & 1]; which body ran last is on the device -- n_pAp -- not on the host)
    const bool fp = fusep_plan(s);
    hipLaunchKernelGGL(cg_x_finalize, dim3(stream_grid(s->nr, 256)), dim3(256), 0, g.stream, s->nr, s->x, fp ? s->pbuf[0] : s->p,
        fp ? s->pbuf[1] : s->p, s->S);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(&s->S->x_pending, 0, sizeof(int), g.stream));
  }
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
#ifdef SB_LAB
  {
    VPhase vp;
    Lead ld[2];
    HIP_CHECK(hipMemcpy(&vp, s->vphase, sizeof vp, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(ld, s->lead, sizeof ld, hipMemcpyDeviceToHost));
    if (ld[0].error || ld[1].error)
      SB_FATAL("rank %d: a lead kernel's workgroups timed out waiting for workgroup 0's scalar step (sb_cg_set_fused(s, 1) / "
               "bench.py --fused 1 selects the separate launches)", g.rank);
    if (vp.error)
      SB_FATAL("rank %d: the one-launch vector phase timed out waiting for its own workgroups: the GPU is shared with other "
               "work (set SB_SHARED_GPU=1 or SB_VPHASE=0 to use the separate launches)", g.rank);
  }
#endif
  // (a rank that fails poisons what it would have published, so the others leave their waits at once with code 2: the rank
  //  that saw the CAUSE -- code 1 or 3 -- is the one whose message matters)
  if (s->halo && s->halo->p2p) {
    int e = 0;
    HIP_CHECK(hipMemcpy(&e, s->halo->err, sizeof e, hipMemcpyDeviceToHost));
    if (e == 1)
      SB_FATAL("rank %d: a neighbour's halo block did not arrive within %lld ms (SB_P2P_TIMEOUT_MS raises the bound, "
               "SB_P2P_HALO=0 selects RCCL)", g.rank, s->halo->push.timeoutTicks / P2P_TICKS_PER_MS);
    if (e && h.p2p_error != 1) h.p2p_error = 2;
  }
  if (h.p2p_error == 2)
    SB_FATAL("rank %d: another rank reported a communication failure over the peer-mapped paths and ended the exchange "
             "(its own message says which wait ran out); this rank stopped with it", g.rank);
  if (h.p2p_error)
    SB_FATAL("rank %d: a peer's contribution to an in-kernel all-reduce did not arrive within %lld ms "
             "(SB_P2P_TIMEOUT_MS raises the bound, SB_P2P=0 selects the RCCL all-reduce)", g.rank,
        g.p2pTimeoutTicks / P2P_TICKS_PER_MS);
  if (s->timing) {
    for (double& v : s->region_ms) v = 0.0;
    for (size_t i = 1; i < s->evUsed; i++) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, s->ev[i - 1], s->ev[i]));
      if (s->evRegion[i] >= 0) s->region_ms[s->evRegion[i]] += ms;
    }
  }
  s->timing       = false;
  s->fusepLatched = -1; // the solve is over: the next sb_cg_start decides anew
  return h.iters + 1; // the value of k when the reference's for loop exits (:107,:140)
}

int sb_cg_solve(sb_cg* s, int itermax, double eps)
{
  sb_cg_start(s, itermax, eps);
  HIP_CHECK(hipEventRecord(s->evLoop0, g.stream));
  sb_cg_run_iters(s, itermax > 1 ? itermax - 1 : 0);
  HIP_CHECK(hipEventRecord(s->evLoop1, g.stream));
  const int k = sb_cg_finish(s);
  HIP_CHECK(hipEventElapsedTime(&s->loop_ms, s->evLoop0, s->evLoop1));
  return k;
}

int sb_cg_history(const sb_cg* s, double* rr_out, int rr_cap, double* pAp_out, int pAp_cap, int* n_pAp)
{
  need_init();
  HIP_CHECK(hipStreamSynchronize(g.stream));
  CgScalars h;
  HIP_CHECK(hipMemcpy(&h, s->S, sizeof h, hipMemcpyDeviceToHost));
  int nrr = h.n_rr < s->hist_cap ? h.n_rr : s->hist_cap;
  int npa = h.n_pAp < s->hist_cap ? h.n_pAp : s->hist_cap;
  if (nrr > rr_cap) nrr = rr_cap;
  if (npa > pAp_cap) npa = pAp_cap;
  if (nrr > 0) HIP_CHECK(hipMemcpy(rr_out, s->rr_hist, (size_t)nrr * sizeof(double), hipMemcpyDeviceToHost));
  if (npa > 0) HIP_CHECK(hipMemcpy(pAp_out, s->pAp_hist, (size_t)npa * sizeof(double), hipMemcpyDeviceToHost));
  if (n_pAp) *n_pAp = npa;
  return nrr;
}

void sb_cg_solution(const sb_cg* s, double* x_host)
{
  need_init();
  double* tmp = scratch_ws(1, s->nr);
  sb_unpermute(s->A, s->x, tmp);
  sb_d2h(x_host, tmp, (size_t)s->nr * sizeof(double));
}

double sb_cg_check_residual(const sb_cg* s)
{
  need_init();
  if (!s->xexact || s->nr == 0) return 0.0;
  const uint32_t blocks = stream_grid(s->nr, 256);
  double* q             = scratch_partials(blocks);
  hipLaunchKernelGGL(max_abs_diff_partials, dim3(blocks), dim3(256), 0, g.stream, s->nr, s->x, s->xexact, q);
  HIP_CHECK(hipGetLastError());
  std::vector<double> h(blocks);
  sb_d2h(h.data(), q, blocks * sizeof(double));
  double m = 0.0;
  for (double v : h)
    if (v > m) m = v;
  if (multi_rank()) { // commReduction(&residual, MAX), src/CGSolver.c:55
    sb_h2d(g.scalar, &m, sizeof m);
    sb_comm_reduction(g.scalar, 0);
    sb_d2h(&m, g.scalar, sizeof m);
  }
  return m;
}

double sb_debug_stream_read_gbs(size_t bytes, int reps)
{ // raw read ceiling of this device: `reps` passes over a `bytes`-sized buffer
  need_init();
  double2* buf = nullptr;
  HIP_CHECK(hipMalloc(&buf, bytes));
  HIP_CHECK(hipMemsetAsync(buf, 0, bytes, g.stream));
  const size_t n2 = bytes / sizeof(double2);
  dim3 grid((unsigned)g.prop.multiProcessorCount * 8), block(256);
  hipLaunchKernelGGL(stream_read_k, grid, block, 0, g.stream, buf, n2, g.scalar);
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(a, g.stream));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(stream_read_k, grid, block, 0, g.stream, buf, n2, g.scalar);
  HIP_CHECK(hipEventRecord(b, g.stream));
  HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  HIP_CHECK(hipEventDestroy(a));
  HIP_CHECK(hipEventDestroy(b));
  HIP_CHECK(hipFree(buf));
  return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

double sb_cg_loop_ms(const sb_cg* s) { return (double)s->loop_ms; }

void sb_cg_region_ms(const sb_cg* s, double out[4])
{
  for (int i = 0; i < 4; i++) out[i] = s->region_ms[i];
}
