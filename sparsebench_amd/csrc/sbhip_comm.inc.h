// sbhip_comm.inc.h -- part of the single translation unit sbhip.hip (textual include, shares its
// static context): communicator (RCCL / host transport), in-kernel all-reduce set-up, halo exchange.
// ===========================================================================
// communicator + halo
// ===========================================================================
void sb_comm_unique_id(void* id_out)
{
  rccl_open();
  ncclUniqueId id;
  RCCL_CHECK(rccl.GetUniqueId(&id));
  memcpy(id_out, &id, SB_UNIQUE_ID_BYTES);
}

void sb_comm_init(int rank, int size, const void* idbytes)
{
  need_init();
  if (g.comm) SB_FATAL("communicator already initialised");
  if (size < 1 || rank < 0 || rank >= size) SB_FATAL("bad rank %d / size %d", rank, size);
  g.rank = rank, g.size = size;
  // serial: every comm call degrades to a no-op (src/comm.c:404-411).  SB_FORCE_RCCL=1
  // builds a 1-rank RCCL communicator anyway, so the multi-rank kernel sequence and the
  // RCCL bindings can be exercised on a single GPU.
  if (size == 1 && !getenv("SB_FORCE_RCCL")) return;
  rccl_open();
  ncclUniqueId id;
  memcpy(&id, idbytes, SB_UNIQUE_ID_BYTES);
  RCCL_CHECK(rccl.CommInitRank(&g.comm, size, id, rank));
  if (size > 1) { // peer-mapped buffers for the in-kernel all-reduce; the handles travel over RCCL
    unsigned char mine[SB_P2P_HANDLE_BYTES], all[SB_P2P_HANDLE_BYTES * P2P_MAX];
    const int have = size <= P2P_MAX && sb_comm_p2p_handle(mine);
    if (!have) memset(mine, 0, sizeof mine);
    if (size <= P2P_MAX) {
      sb_comm_allgather_bytes(mine, SB_P2P_HANDLE_BYTES, all);
      sb_comm_p2p_open(have ? all : nullptr);
    }
  }
}

void sb_comm_init_transport(int rank, int size, const sb_transport* t)
{
  need_init();
  if (g.comm || g.hasXport) SB_FATAL("communicator already initialised");
  if (size < 1 || rank < 0 || rank >= size || !t || !t->allreduce || !t->neighbour_exchange)
    SB_FATAL("bad transport / rank %d / size %d", rank, size);
  g.rank = rank, g.size = size, g.xport = *t, g.hasXport = true;
}

// ---- in-kernel all-reduce over peer-mapped memory: set-up ---------------------------------
static void p2p_release()
{
  for (int r = 0; r < P2P_MAX; r++) {
    if (g.p2pPeer[r] && g.p2pPeer[r] != (void*)g.p2pBuf) (void)hipIpcCloseMemHandle(g.p2pPeer[r]);
    g.p2pPeer[r] = nullptr;
  }
  if (g.p2pBuf) (void)hipFree(g.p2pBuf);
  if (g.p2pView) (void)hipFree(g.p2pView);
  g.p2pBuf = nullptr, g.p2pView = nullptr, g.p2pOn = false, g.p2pSeq = 0;
}

int sb_comm_p2p_handle(unsigned char* handle_out)
{
  need_init();
  const char* env = getenv("SB_P2P");
  if (env && atoi(env) == 0) {
    snprintf(g.p2pReason, sizeof g.p2pReason, "off: SB_P2P=0");
    return 0;
  }
  // test hook: SB_P2P_FAIL_RANK=r makes rank r behave as if its buffer could not be exported, to
  // exercise the collective fall-back decision (tests/test_gpu_multirank.py)
  const char* failRank = getenv("SB_P2P_FAIL_RANK");
  if (failRank && atoi(failRank) == g.rank) {
    snprintf(g.p2pReason, sizeof g.p2pReason, "off: SB_P2P_FAIL_RANK=%d (test hook)", g.rank);
    return 0;
  }
  static_assert(sizeof(hipIpcMemHandle_t) <= SB_P2P_HANDLE_BYTES, "IPC handle size");
  if (!g.p2pBuf) {
    void* buf = nullptr; // fine-grained: coherent between GPUs while kernels are running
    const hipError_t e = hipExtMallocWithFlags(&buf, 2 * P2P_MAX * sizeof(P2PSlot), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      snprintf(g.p2pReason, sizeof g.p2pReason, "off: fine-grained allocation failed on rank %d (%s)", g.rank, hipGetErrorName(e));
      return 0;
    }
    g.p2pBuf = (P2PSlot*)buf;
    HIP_CHECK(hipMemset(g.p2pBuf, 0, 2 * P2P_MAX * sizeof(P2PSlot)));
    HIP_CHECK(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  const hipError_t eh = hipIpcGetMemHandle(&h, g.p2pBuf);
  if (eh != hipSuccess) {
    (void)hipGetLastError();
    p2p_release();
    snprintf(g.p2pReason, sizeof g.p2pReason, "off: hipIpcGetMemHandle failed on rank %d (%s)", g.rank, hipGetErrorName(eh));
    return 0;
  }
  memset(handle_out, 0, SB_P2P_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof h);
  return 1;
}

int sb_comm_p2p_open(const unsigned char* all_handles)
{
  need_init();
  if (!multi_rank()) return 0;
  { // waits inside CG: generous, configurable; a late peer is not a dead peer
    const char* t = getenv("SB_P2P_TIMEOUT_MS");
    const long long ms = t && atoll(t) > 0 ? atoll(t) : 30000;
    g.p2pTimeoutTicks = ms * P2P_TICKS_PER_MS;
  }
  char why[200] = "";
  int ok = all_handles != nullptr && g.p2pBuf != nullptr && g.size <= P2P_MAX;
  if (!ok) snprintf(why, sizeof why, "rank %d: %s", g.rank, g.size > P2P_MAX ? "more than 16 ranks" : g.p2pReason);
  unsigned char zero[SB_P2P_HANDLE_BYTES] = { 0 };
  for (int r = 0; ok && r < g.size; r++) {
    const unsigned char* hb = all_handles + (size_t)r * SB_P2P_HANDLE_BYTES;
    if (memcmp(hb, zero, SB_P2P_HANDLE_BYTES) == 0) {
      ok = 0; // that rank has none
      snprintf(why, sizeof why, "rank %d exported no buffer", r);
    } else if (r == g.rank) g.p2pPeer[r] = g.p2pBuf;
    else {
      hipIpcMemHandle_t h;
      memcpy(&h, hb, sizeof h);
      const hipError_t e = hipIpcOpenMemHandle(&g.p2pPeer[r], h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        g.p2pPeer[r] = nullptr;
        ok = 0;
        snprintf(why, sizeof why, "rank %d: hipIpcOpenMemHandle of rank %d's buffer failed (%s)", g.rank, r, hipGetErrorName(e));
      }
    }
  }
  // every rank must come to the same decision.  Round 1 (the established transport): did everybody
  // map everybody?  Round 2: SIX in-kernel exchanges (three per slot parity) with a value that changes
  // every time -- a stale line of an earlier exchange of the same parity would be caught -- checked,
  // and agreed on over the transport again.
  double* d = (double*)sb_malloc(4 * sizeof(double));
  auto agree = [&](int mine) {
    const double v = mine ? 1.0 : 0.0;
    sb_h2d(d, &v, sizeof v);
    sb_comm_reduction(d, 1);
    double sum = 0.0;
    sb_d2h(&sum, d, sizeof sum);
    return sum == (double)g.size;
  };
  // (the first launch of a kernel of this library loads its code object, which takes a rank-dependent while: do that
  //  before the ranks synchronise, so that the bounded waits of the self-test start within microseconds of each other)
  hipLaunchKernelGGL(gather_k, dim3(1), dim3(64), 0, g.stream, 0u, (const uint32_t*)nullptr, (const double*)nullptr,
      (double*)nullptr, (const int*)nullptr);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  bool on = agree(ok);
  if (!on && ok) snprintf(why, sizeof why, "another rank could not export / map a buffer");
  double p2pUs = 0.0;
  if (on) {
    P2PView view;
    memset(&view, 0, sizeof view);
    view.rank = g.rank, view.size = g.size;
    view.timeoutTicks = 5000 * P2P_TICKS_PER_MS; // self-test: 5 s
    for (int r = 0; r < g.size; r++) view.peer[r] = (P2PSlot*)g.p2pPeer[r];
    HIP_CHECK(hipMalloc(&g.p2pView, sizeof view));
    HIP_CHECK(hipMemcpy(g.p2pView, &view, sizeof view, hipMemcpyHostToDevice));
    int* err = (int*)(d + 2);
    int good = 1;
    const int rounds = 6;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int it = 0; it < rounds && good; it++) {
      HIP_CHECK(hipMemsetAsync(d, 0, 4 * sizeof(double), g.stream));
      if (it == 2) HIP_CHECK(hipEventRecord(e0, g.stream));
      const double mine = (double)(g.rank + 1) * (double)(it + 1) + 0.25 * it;
      hipLaunchKernelGGL(p2p_selftest_k, dim3(1), dim3(64), 0, g.stream, (const P2PView*)g.p2pView, ++g.p2pSeq, mine,
          d + 1, err);
      HIP_CHECK(hipGetLastError());
      if (it == rounds - 1) HIP_CHECK(hipEventRecord(e1, g.stream));
      double got = 0.0;
      int e      = 0;
      sb_d2h(&got, d + 1, sizeof got);
      sb_d2h(&e, err, sizeof e);
      double want = 0.0; // the same pairwise tree the kernel uses
      {
        double v[P2P_MAX];
        int n = g.size;
        for (int r = 0; r < n; r++) v[r] = (double)(r + 1) * (double)(it + 1) + 0.25 * it;
        while (n > 1) {
          const int h = n >> 1;
          for (int i = 0; i < h; i++) v[i] = v[2 * i] + v[2 * i + 1];
          if (n & 1) v[h] = v[n - 1];
          n = h + (n & 1);
        }
        want = v[0];
      }
      if (e || got != want) {
        good = 0;
        snprintf(why, sizeof why, "rank %d: self-test exchange %d %s", g.rank, it, e ? "timed out (5 s)" : "returned a wrong sum");
      }
    }
    float ms = 0.f;
    if (good && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) p2pUs = 1e3 * ms / (rounds - 2);
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    on = agree(good);
    if (!on && good) snprintf(why, sizeof why, "the self-test failed on another rank");
    if (on) { // production bound of the waits
      view.timeoutTicks = g.p2pTimeoutTicks;
      HIP_CHECK(hipMemcpy(g.p2pView, &view, sizeof view, hipMemcpyHostToDevice));
    }
  }
  sb_free(d);
  if (!on) p2p_release();
  g.p2pOn = on;
  if (on) snprintf(g.p2pReason, sizeof g.p2pReason, "on: %d ranks mapped, 6 self-test exchanges ok (~%.1f us each incl. launch + readback)", g.size, p2pUs);
  else snprintf(g.p2pReason, sizeof g.p2pReason, "off: %s", why[0] ? why : "unknown");
  if (getenv("SB_PACK_REPORT") || getenv("SB_P2P_REPORT"))
    fprintf(stderr, "sbhip comm: rank %d/%d in-kernel all-reduce over peer-mapped memory: %s\n", g.rank, g.size, g.p2pReason);
  return on ? 1 : 0;
}

int sb_comm_p2p_enabled(void) { return g.p2pOn ? 1 : 0; }
const char* sb_comm_p2p_reason(void) { return g.p2pReason; }

// Which data plane the CG loop uses from now on: 1 (default) the peer-mapped paths where their set-up succeeded, 0 the
// communicator's own collectives (all-reduce, send / recv) although the mappings exist -- what SB_P2P=0 SB_P2P_HALO=0
// would have given, without tearing anything down, so that one process can time both (bench.py).  Collective by
// contract: every rank calls it with the same value, between solves (the exchange counters of both paths stay in
// step on all ranks because every rank switches at the same point of its call sequence).
void sb_comm_data_plane(int peer_mapped)
{
  need_init();
  HIP_CHECK(hipStreamSynchronize(g.stream));
  g.p2pUse = peer_mapped != 0;
}
int sb_comm_data_plane_selected(void) { return g.p2pUse ? 1 : 0; }

// Peer-mapped halo exchange, variant: 1 = the rank's halo push rides in the SpMV launch (its first 16 workgroups send
// p[elementsToSend]; the tiles that wait for the neighbours' pushes are stored last) instead of a launch of its own --
// one launch fewer per loop body; 0 (default; SB_HALO_PUSH_INSIDE=1 changes the default) = separate halo_push_k.
// Bit-identical either way.  Which one is faster can only be decided with one rank per GPU (ranks sharing a GPU keep
// each other's pushes off the CUs), so bench.py times both on a real node.  Collective, between solves.
void sb_comm_halo_push_inside(int on)
{
  need_init();
  HIP_CHECK(hipStreamSynchronize(g.stream));
  g.pushInside = on != 0;
}

// what the RCCL communicator itself reports: out = {ranks in the communicator, this rank's id in it, HIP device it is
// bound to}; returns 1, or 0 when there is no RCCL communicator (one rank, host transport) or the library lacks the queries
int sb_comm_rccl_info(int out[3])
{
  out[0] = out[1] = out[2] = -1;
  if (!g.comm || !rccl.CommCount || !rccl.CommUserRank || !rccl.CommCuDevice) return 0;
  RCCL_CHECK(rccl.CommCount(g.comm, &out[0]));
  RCCL_CHECK(rccl.CommUserRank(g.comm, &out[1]));
  RCCL_CHECK(rccl.CommCuDevice(g.comm, &out[2]));
  return 1;
}

void sb_comm_finalize(void)
{
  if (g.init) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    p2p_release();
  }
  g.hasXport = false;
  if (g.comm) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    RCCL_CHECK(rccl.CommDestroy(g.comm));
    g.comm = nullptr;
  }
  g.rank = 0, g.size = 1;
}

int sb_comm_rank(void) { return g.rank; }
int sb_comm_size(void) { return g.size; }

void sb_comm_reduction(double* v_dev, int op)
{
  need_init();
  if (g.hasXport) {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    g.xport.allreduce(g.xport.ctx, v_dev, op);
    return;
  }
  if (!g.comm) return;
  RCCL_CHECK(rccl.AllReduce(v_dev, v_dev, 1, ncclFloat64_, op == 0 ? ncclMax_ : ncclSum_, g.comm,
      g.stream));
}

void sb_comm_allgather_bytes(const void* mine_host, int nbytes, void* all_host)
{
  need_init();
  if (!g.comm) {
    memcpy(all_host, mine_host, (size_t)nbytes);
    return;
  }
  char *dsend = nullptr, *drecv = nullptr;
  HIP_CHECK(hipMalloc(&dsend, (size_t)nbytes + 8));
  HIP_CHECK(hipMalloc(&drecv, (size_t)nbytes * g.size + 8));
  HIP_CHECK(hipMemcpyAsync(dsend, mine_host, (size_t)nbytes, hipMemcpyHostToDevice, g.stream));
  RCCL_CHECK(rccl.AllGather(dsend, drecv, (size_t)nbytes, ncclInt8_, g.comm, g.stream));
  HIP_CHECK(hipMemcpyAsync(all_host, drecv, (size_t)nbytes * g.size, hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  HIP_CHECK(hipFree(dsend));
  HIP_CHECK(hipFree(drecv));
}

void sb_comm_alltoallv_ints(const int* sendbuf, const int* sendcounts, const int* sdispls, int* recvbuf,
    const int* recvcounts, const int* rdispls)
{
  need_init();
  const int me = g.rank;
  if (!g.comm) {
    memcpy(recvbuf + rdispls[0], sendbuf + sdispls[0], (size_t)sendcounts[0] * sizeof(int));
    return;
  }
  size_t ns = 0, nr = 0;
  for (int r = 0; r < g.size; r++) {
    if ((size_t)(sdispls[r] + sendcounts[r]) > ns) ns = (size_t)(sdispls[r] + sendcounts[r]);
    if ((size_t)(rdispls[r] + recvcounts[r]) > nr) nr = (size_t)(rdispls[r] + recvcounts[r]);
  }
  int *dsend = nullptr, *drecv = nullptr;
  HIP_CHECK(hipMalloc(&dsend, (ns + 2) * sizeof(int)));
  HIP_CHECK(hipMalloc(&drecv, (nr + 2) * sizeof(int)));
  HIP_CHECK(hipMemcpyAsync(dsend, sendbuf, ns * sizeof(int), hipMemcpyHostToDevice, g.stream));
  RCCL_CHECK(rccl.GroupStart());
  for (int r = 0; r < g.size; r++) {
    if (r == me) continue;
    if (sendcounts[r])
      RCCL_CHECK(rccl.Send(dsend + sdispls[r], (size_t)sendcounts[r], ncclInt32_, r, g.comm, g.stream));
    if (recvcounts[r])
      RCCL_CHECK(rccl.Recv(drecv + rdispls[r], (size_t)recvcounts[r], ncclInt32_, r, g.comm, g.stream));
  }
  RCCL_CHECK(rccl.GroupEnd());
  if (sendcounts[me])
    HIP_CHECK(hipMemcpyAsync(drecv + rdispls[me], dsend + sdispls[me], (size_t)sendcounts[me] * sizeof(int),
        hipMemcpyDeviceToDevice, g.stream));
  HIP_CHECK(hipMemcpyAsync(recvbuf, drecv, nr * sizeof(int), hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  HIP_CHECK(hipFree(dsend));
  HIP_CHECK(hipFree(drecv));
}

void sb_comm_barrier(void)
{
  need_init();
  if (g.comm) {
    HIP_CHECK(hipMemsetAsync(g.scalar, 0, sizeof(double), g.stream));
    sb_comm_reduction(g.scalar, 1);
  }
  HIP_CHECK(hipStreamSynchronize(g.stream));
}

// ---- halo exchange over peer-mapped memory: collective set-up (every rank calls sb_halo_create) -------
static void halo_exchange(sb_halo* h, double* x, const int* stop, hipStream_t stream = nullptr, bool inCG = false,
    bool pushOnly = false);

static void halo_p2p_release(sb_halo* h)
{
  for (int i = 0; i < P2P_MAX; i++) {
    if (h->peerStage[i]) (void)hipIpcCloseMemHandle(h->peerStage[i]);
    h->peerStage[i] = nullptr;
  }
  if (h->stage) (void)hipFree(h->stage);
  sb_free(h->slot), sb_free(h->dest), sb_free(h->done), sb_free(h->dSrcRank), sb_free(h->dRdispl), sb_free(h->dRcount);
  sb_free(h->err), sb_free(h->dPush);
  h->dPush = nullptr;
  h->stage = nullptr, h->slot = nullptr, h->dest = nullptr, h->done = nullptr, h->dSrcRank = nullptr;
  h->dRdispl = nullptr, h->dRcount = nullptr, h->err = nullptr, h->p2p = false;
}

static bool comm_can_allgather() { return g.comm != nullptr || (g.hasXport && g.xport.allgather_bytes != nullptr); }
static void comm_allgather(const void* mine, int nbytes, void* all)
{
  if (g.comm) sb_comm_allgather_bytes(mine, nbytes, all);
  else g.xport.allgather_bytes(g.xport.ctx, mine, nbytes, all);
}

static void halo_p2p_setup(sb_halo* h)
{
  const char* env = getenv("SB_P2P_HALO");
  // rides on the in-kernel all-reduce's decision (same memory model, same agreement); needs an
  // all-gather for the handles.  Every condition here is the same on every rank.
  if (!g.p2pOn || g.size > P2P_MAX || !comm_can_allgather() || (env && atoi(env) == 0)) {
    snprintf(h->p2pReason, sizeof h->p2pReason, "off: %s", (env && atoi(env) == 0) ? "SB_P2P_HALO=0"
        : !g.p2pOn ? "the in-kernel all-reduce is off (same memory model)" : g.size > P2P_MAX ? "more than 16 ranks"
        : "the transport has no all-gather for the handles");
    return;
  }
  const int P = g.size, E = h->externalCount;
  int ok      = 1;
  char why[200] = "";
  // 1. own staging area + flags, exported
  struct Info {
    unsigned char handle[SB_P2P_HANDLE_BYTES];
    int ext;
    int rdisplOf[P2P_MAX]; // where rank r's block starts in my area, -1: r sends me nothing
  } mine, all[P2P_MAX];
  memset(&mine, 0, sizeof mine);
  mine.ext = E;
  for (int r = 0; r < P2P_MAX; r++) mine.rdisplOf[r] = -1;
  for (int j = 0; j < h->indegree; j++) mine.rdisplOf[h->sources[j]] = h->rdispls[j];
  const size_t words = 2 * (size_t)E + 2 * P2P_MAX + 8;
  void* buf          = nullptr;
  if (hipExtMallocWithFlags(&buf, words * sizeof(unsigned long long), hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    ok = 0;
    snprintf(why, sizeof why, "rank %d: fine-grained allocation of the staging area failed", g.rank);
  } else {
    h->stage = (unsigned long long*)buf;
    HIP_CHECK(hipMemset(h->stage, 0, words * sizeof(unsigned long long)));
    HIP_CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t ih;
    if (hipIpcGetMemHandle(&ih, h->stage) != hipSuccess) {
      (void)hipGetLastError();
      ok = 0;
      snprintf(why, sizeof why, "rank %d: hipIpcGetMemHandle of the staging area failed", g.rank);
    } else memcpy(mine.handle, &ih, sizeof ih);
  }
  comm_allgather(&mine, (int)sizeof mine, all);
  // 2. map the destinations, work out where every sent element lands
  memset(&h->push, 0, sizeof h->push);
  std::vector<uint32_t> slot((size_t)h->totalSend);
  std::vector<uint8_t> dest((size_t)h->totalSend);
  unsigned char zero[SB_P2P_HANDLE_BYTES] = { 0 };
  for (int i = 0; ok && i < h->outdegree; i++) {
    const int d = h->destinations[i];
    if (i >= P2P_MAX || memcmp(all[d].handle, zero, sizeof zero) == 0 || all[d].rdisplOf[g.rank] < 0) {
      ok = 0;
      snprintf(why, sizeof why, "rank %d: destination %d exported no staging area / does not expect this rank", g.rank, d);
      break;
    }
    hipIpcMemHandle_t ih;
    memcpy(&ih, all[d].handle, sizeof ih);
    if (hipIpcOpenMemHandle(&h->peerStage[i], ih, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      h->peerStage[i] = nullptr;
      ok = 0;
      snprintf(why, sizeof why, "rank %d: hipIpcOpenMemHandle of rank %d's staging area failed", g.rank, d);
      break;
    }
    h->push.stage[i] = (unsigned long long*)h->peerStage[i];
    h->push.flag[i]  = (unsigned long long*)h->peerStage[i] + 2 * (size_t)all[d].ext;
    h->push.ext[i]   = (uint32_t)all[d].ext;
    for (int e = 0; e < h->sendCounts[i]; e++) {
      slot[(size_t)h->sdispls[i] + e] = (uint32_t)(all[d].rdisplOf[g.rank] + e);
      dest[(size_t)h->sdispls[i] + e] = (uint8_t)i;
    }
  }
  // 3. every rank must come to the same decision, before and after one tested exchange
  double* dflag = (double*)sb_malloc(sizeof(double));
  auto agree = [&](int good) {
    const double v = good ? 1.0 : 0.0;
    sb_h2d(dflag, &v, sizeof v);
    sb_comm_reduction(dflag, 1);
    double sum = 0.0;
    sb_d2h(&sum, dflag, sizeof sum);
    return sum == (double)P;
  };
  bool on = agree(ok);
  if (!on && ok) snprintf(why, sizeof why, "another rank could not export / map a staging area");
  if (on) {
    // TEST HOOK (bench.py's degraded completion, tests/test_gpu_bench.py): SB_TEST_CORRUPT_P2P_HALO=r makes rank r's
    // peer-mapped push deliver its first two values in each other's places -- wrong on the peer-mapped plane ONLY (the
    // communicator's send / recv does not use the slot list; the self-test below sends one value in every slot)
    if (const char* bad = getenv("SB_TEST_CORRUPT_P2P_HALO")) {
      if (atoi(bad) == g.rank && h->totalSend >= 2 && dest[0] == dest[1]) {
        std::swap(slot[0], slot[1]);
        fprintf(stderr, "sbhip: rank %d: SB_TEST_CORRUPT_P2P_HALO is set: the peer-mapped push swaps its first two values (test hook)\n", g.rank);
      }
    }
    h->slot = (uint32_t*)upload(slot.data(), slot.size() * sizeof(uint32_t));
    h->dest = (uint8_t*)upload(dest.data(), dest.size());
    h->done = (unsigned int*)sb_malloc(sizeof(unsigned int));
    h->err  = (int*)sb_malloc(sizeof(int));
    HIP_CHECK(hipMemset(h->done, 0, sizeof(unsigned int)));
    HIP_CHECK(hipMemset(h->err, 0, sizeof(int)));
    h->dSrcRank = (int*)upload(h->sources.data(), h->sources.size() * sizeof(int));
    h->dRdispl  = (int*)upload(h->rdispls.data(), h->rdispls.size() * sizeof(int));
    h->dRcount  = (int*)upload(h->recvCounts.data(), h->recvCounts.size() * sizeof(int));
    h->push.n = (uint32_t)h->totalSend, h->push.ndest = h->outdegree, h->push.rank = g.rank;
    h->push.packIdx = h->packIdx, h->push.slot = h->slot, h->push.dest = h->dest, h->push.done = h->done;
    h->push.err = h->err, h->push.p2pErr = nullptr; // (the CG loop adds its control block's flag: sbhip_cg.inc.h)
    h->push.timeoutTicks = 5000 * P2P_TICKS_PER_MS; // self-test: 5 s
    // self-test: SIX exchanges (three per parity of the alternating staging areas); in exchange `it` every rank
    // sends value(rank, it) in all its slots, so a block that still holds an earlier exchange's data is caught
    const size_t nvec = (size_t)h->nr + (size_t)E + 1;
    double* v         = (double*)sb_malloc(nvec * sizeof(double));
    std::vector<double> host(nvec);
    auto value = [](int rank, int it) { return (double)(rank + 1) + 1000.0 * (double)it; };
    h->p2p   = true;
    int good = 1;
    for (int it = 0; it < 6 && good; it++) {
      std::fill(host.begin(), host.end(), value(g.rank, it));
      sb_h2d(v, host.data(), nvec * sizeof(double));
      halo_exchange(h, v, nullptr, g.stream, true);
      sb_d2h(host.data(), v, nvec * sizeof(double));
      int e = 0;
      sb_d2h(&e, h->err, sizeof e);
      if (e) {
        good = 0;
        snprintf(why, sizeof why, "rank %d: self-test exchange %d timed out (5 s)", g.rank, it);
      }
      for (int j = 0; good && j < h->indegree; j++)
        for (int i = 0; i < h->recvCounts[j]; i++)
          if (host[(size_t)h->nr + h->rdispls[j] + i] != value(h->sources[j], it)) {
            good = 0;
            snprintf(why, sizeof why, "rank %d: self-test exchange %d delivered wrong data from rank %d", g.rank, it, h->sources[j]);
            break;
          }
    }
    sb_free(v);
    on = agree(good);
    if (!on && good) snprintf(why, sizeof why, "the self-test failed on another rank");
    h->push.timeoutTicks = g.p2pTimeoutTicks;
    // TEST HOOKS (tests/test_gpu_bench.py: a failure on one rank must end the others at once): SB_TEST_DROP_PUSH_RANK=r with
    // SB_TEST_DROP_PUSH_AT=k makes rank r "forget" to announce its k-th halo exchange from now on; SB_TEST_HALO_TIMEOUT_MS bounds
    // the halo waits separately from the all-reduce's (SB_P2P_TIMEOUT_MS), so that the test can tell who timed out
    if (const char* dr = getenv("SB_TEST_DROP_PUSH_RANK")) {
      const char* at = getenv("SB_TEST_DROP_PUSH_AT");
      if (atoi(dr) == g.rank && at && atoll(at) > 0) {
        h->push.dropSeq = h->seq + (unsigned long long)atoll(at);
        fprintf(stderr, "sbhip: rank %d: SB_TEST_DROP_PUSH_RANK is set: halo exchange %llu will not be announced (test hook)\n", g.rank,
            h->push.dropSeq);
      }
    }
    if (const char* ht = getenv("SB_TEST_HALO_TIMEOUT_MS"))
      if (atoll(ht) > 0) h->push.timeoutTicks = atoll(ht) * P2P_TICKS_PER_MS;
  }
  sb_free(dflag);
  if (!on) halo_p2p_release(h);
  h->p2p = on;
  if (on) h->dPush = (HaloPush*)upload(&h->push, sizeof h->push);
  if (on) snprintf(h->p2pReason, sizeof h->p2pReason, "on: %d destinations mapped, 6 self-test exchanges ok", h->outdegree);
  else snprintf(h->p2pReason, sizeof h->p2pReason, "off: %s", why[0] ? why : "unknown");
  if (getenv("SB_PACK_REPORT") || getenv("SB_P2P_REPORT"))
    fprintf(stderr, "sbhip comm: rank %d/%d halo exchange over peer-mapped memory: %s\n", g.rank, g.size, h->p2pReason);
}

sb_halo* sb_halo_create(uint32_t nr, int outdegree, const int* destinations, const int* sendCounts,
    const int* sdispls, int indegree, const int* sources, const int* recvCounts, const int* rdispls,
    const int* elementsToSend, int totalSendCount, int externalCount, const uint32_t* oldToNewPerm)
{
  need_init();
  sb_halo* h        = new sb_halo();
  h->nr             = nr;
  h->outdegree      = outdegree;
  h->indegree       = indegree;
  h->totalSend      = totalSendCount;
  h->externalCount  = externalCount;
  h->destinations.assign(destinations, destinations + outdegree);
  h->sendCounts.assign(sendCounts, sendCounts + outdegree);
  h->sdispls.assign(sdispls, sdispls + outdegree);
  h->sources.assign(sources, sources + indegree);
  h->recvCounts.assign(recvCounts, recvCounts + indegree);
  h->rdispls.assign(rdispls, rdispls + indegree);
  int sum = 0;
  for (int i = 0; i < outdegree; i++) {
    if (sdispls[i] != sum) SB_FATAL("halo: sdispls must be the prefix sums of sendCounts");
    sum += sendCounts[i];
  }
  if (sum != totalSendCount) SB_FATAL("halo: totalSendCount mismatch");
  sum = 0;
  for (int i = 0; i < indegree; i++) {
    if (rdispls[i] != sum) SB_FATAL("halo: rdispls must be the prefix sums of recvCounts");
    sum += recvCounts[i];
  }
  if (sum != externalCount) SB_FATAL("halo: externalCount mismatch");
  std::vector<uint32_t> idx((size_t)totalSendCount);
  for (int i = 0; i < totalSendCount; i++) {
    if (elementsToSend[i] < 0 || (uint32_t)elementsToSend[i] >= nr)
      SB_FATAL("halo: elementsToSend[%d]=%d out of range", i, elementsToSend[i]);
    idx[i] = oldToNewPerm ? oldToNewPerm[elementsToSend[i]] : (uint32_t)elementsToSend[i];
  }
  // TEST HOOK (bench.py's pre-flight gate, tests/test_gpu_bench.py): SB_TEST_CORRUPT_HALO=r makes rank r send the
  // value of ANOTHER of its rows in its first halo slot -- one wrong value arriving at one neighbour, on whichever data
  // plane carries the exchange -- to prove that a mis-delivered halo line cannot pass the known-answer checks.
  if (const char* bad = getenv("SB_TEST_CORRUPT_HALO")) {
    if (atoi(bad) == g.rank && totalSendCount >= 4) {
      idx[0] = idx[(size_t)totalSendCount / 2 + 1];
      fprintf(stderr, "sbhip: rank %d: SB_TEST_CORRUPT_HALO is set: halo slot 0 deliberately carries the wrong row (test hook)\n", g.rank);
    }
  }
  h->packIdx = (uint32_t*)upload(idx.data(), idx.size() * sizeof(uint32_t));
  HIP_CHECK(hipMalloc(&h->sendBuf, ((size_t)totalSendCount + 1) * sizeof(double)));
  halo_p2p_setup(h);
  return h;
}

int sb_halo_p2p_enabled(const sb_halo* h) { return h && h->p2p ? 1 : 0; }
// set up AND selected (sb_comm_data_plane)
static inline bool halo_p2p_active(const sb_halo* h) { return h && h->p2p && g.p2pUse; }
const char* sb_halo_p2p_reason(const sb_halo* h) { return h ? h->p2pReason : "no halo plan (one rank)"; }

void sb_halo_free(sb_halo* h)
{
  if (!h) return;
  sb_free(h->packIdx), sb_free(h->sendBuf);
  halo_p2p_release(h);
  delete h;
}

// inCG: the caller guarantees an all-reduce between any two exchanges (the CG loop: two per
// iteration), which is what lets the peer-mapped path alternate between just two staging areas.
// pushOnly: the consumer (pattern SpMV, HALO instantiation) waits for the flags and reads the staging area itself.
static void halo_exchange(sb_halo* h, double* x, const int* stop, hipStream_t stream, bool inCG, bool pushOnly)
{
  if (!h || g.size == 1) return;
  if (!stream) stream = g.stream;
  if (halo_p2p_active(h) && inCG) { // push into the neighbours' staging areas, pull the own one into the tail of x
    const unsigned long long seq = ++h->seq;
    if (h->totalSend)
      hipLaunchKernelGGL(halo_push_k, dim3(stream_grid(h->totalSend, 256)), dim3(256), 0, stream, h->push, x, seq, stop);
    if (h->indegree && !pushOnly)
      hipLaunchKernelGGL(halo_pull_k, dim3(h->indegree), dim3(256), 0, stream, h->dSrcRank, h->dRdispl, h->dRcount,
          h->stage, h->stage + 2 * (size_t)h->externalCount, (uint32_t)h->externalCount, x + h->nr, seq, h->err,
          const_cast<int*>(stop), h->push.timeoutTicks);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (h->totalSend) {
    hipLaunchKernelGGL(gather_k, dim3(stream_grid(h->totalSend, 256)), dim3(256), 0, stream,
        (uint32_t)h->totalSend, h->packIdx, x, h->sendBuf, stop);
    HIP_CHECK(hipGetLastError());
  }
  // neighbour all-to-all (MPI_Neighbor_alltoallv, src/comm.c:640-648) as one
  // RCCL group of point-to-point transfers over xGMI, received straight into the
  // tail of x (no unpack), stream-ordered.
  if (g.hasXport) {
    HIP_CHECK(hipStreamSynchronize(stream));
    g.xport.neighbour_exchange(g.xport.ctx, h->sendBuf, h->outdegree, h->destinations.data(),
        h->sendCounts.data(), h->sdispls.data(), x + h->nr, h->indegree, h->sources.data(),
        h->recvCounts.data(), h->rdispls.data());
    return;
  }
  RCCL_CHECK(rccl.GroupStart());
  for (int i = 0; i < h->outdegree; i++)
    RCCL_CHECK(rccl.Send(h->sendBuf + h->sdispls[i], (size_t)h->sendCounts[i], ncclFloat64_,
        h->destinations[i], g.comm, stream));
  for (int i = 0; i < h->indegree; i++)
    RCCL_CHECK(rccl.Recv(x + h->nr + h->rdispls[i], (size_t)h->recvCounts[i], ncclFloat64_,
        h->sources[i], g.comm, stream));
  RCCL_CHECK(rccl.GroupEnd());
}

void sb_halo_exchange(sb_halo* h, double* x)
{
  need_init();
  halo_exchange(h, x, nullptr);
}
