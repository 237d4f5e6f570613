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
  if (env && atoi(env) == 0) return 0;
  // test hook: SB_P2P_FAIL_RANK=r makes rank r behave as if its buffer could not be exported, to
  // exercise the collective fall-back decision (tests/test_gpu_multirank.py)
  const char* failRank = getenv("SB_P2P_FAIL_RANK");
  if (failRank && atoi(failRank) == g.rank) return 0;
  static_assert(sizeof(hipIpcMemHandle_t) <= SB_P2P_HANDLE_BYTES, "IPC handle size");
  if (!g.p2pBuf) {
    void* buf = nullptr; // fine-grained: coherent between GPUs while kernels are running
    if (hipExtMallocWithFlags(&buf, 2 * P2P_MAX * sizeof(P2PSlot), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    g.p2pBuf = (P2PSlot*)buf;
    HIP_CHECK(hipMemset(g.p2pBuf, 0, 2 * P2P_MAX * sizeof(P2PSlot)));
    HIP_CHECK(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, g.p2pBuf) != hipSuccess) {
    (void)hipGetLastError();
    p2p_release();
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
  int ok = all_handles != nullptr && g.p2pBuf != nullptr && g.size <= P2P_MAX;
  unsigned char zero[SB_P2P_HANDLE_BYTES] = { 0 };
  for (int r = 0; ok && r < g.size; r++) {
    const unsigned char* hb = all_handles + (size_t)r * SB_P2P_HANDLE_BYTES;
    if (memcmp(hb, zero, SB_P2P_HANDLE_BYTES) == 0) ok = 0; // that rank has none
    else if (r == g.rank) g.p2pPeer[r] = g.p2pBuf;
    else {
      hipIpcMemHandle_t h;
      memcpy(&h, hb, sizeof h);
      if (hipIpcOpenMemHandle(&g.p2pPeer[r], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        g.p2pPeer[r] = nullptr;
        ok = 0;
      }
    }
  }
  // every rank must come to the same decision.  Round 1 (the established transport): did everybody
  // map everybody?  Round 2: one in-kernel exchange, checked, and agreed on over the transport again.
  double* d = (double*)sb_malloc(4 * sizeof(double));
  auto agree = [&](int mine) {
    const double v = mine ? 1.0 : 0.0;
    sb_h2d(d, &v, sizeof v);
    sb_comm_reduction(d, 1);
    double sum = 0.0;
    sb_d2h(&sum, d, sizeof sum);
    return sum == (double)g.size;
  };
  bool on = agree(ok);
  if (on) {
    P2PView view;
    memset(&view, 0, sizeof view);
    view.rank = g.rank, view.size = g.size;
    for (int r = 0; r < g.size; r++) view.peer[r] = (P2PSlot*)g.p2pPeer[r];
    HIP_CHECK(hipMalloc(&g.p2pView, sizeof view));
    HIP_CHECK(hipMemcpy(g.p2pView, &view, sizeof view, hipMemcpyHostToDevice));
    int* err = (int*)(d + 2);
    HIP_CHECK(hipMemset(d, 0, 4 * sizeof(double)));
    hipLaunchKernelGGL(p2p_selftest_k, dim3(1), dim3(64), 0, g.stream, (const P2PView*)g.p2pView, ++g.p2pSeq,
        (double)(g.rank + 1), d + 1, err);
    HIP_CHECK(hipGetLastError());
    double got = 0.0;
    int e      = 0;
    sb_d2h(&got, d + 1, sizeof got);
    sb_d2h(&e, err, sizeof e);
    on = agree(!e && got == 0.5 * g.size * (g.size + 1));
  }
  sb_free(d);
  if (!on) p2p_release();
  g.p2pOn = on;
  if (getenv("SB_PACK_REPORT") || getenv("SB_P2P_REPORT"))
    fprintf(stderr, "sbhip comm: rank %d/%d in-kernel all-reduce over peer-mapped memory: %s\n", g.rank, g.size,
        on ? "on" : "off (RCCL / transport all-reduce)");
  return on ? 1 : 0;
}

int sb_comm_p2p_enabled(void) { return g.p2pOn ? 1 : 0; }

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
  h->packIdx = (uint32_t*)upload(idx.data(), idx.size() * sizeof(uint32_t));
  HIP_CHECK(hipMalloc(&h->sendBuf, ((size_t)totalSendCount + 1) * sizeof(double)));
  return h;
}

void sb_halo_free(sb_halo* h)
{
  if (!h) return;
  sb_free(h->packIdx), sb_free(h->sendBuf);
  delete h;
}

static void halo_exchange(sb_halo* h, double* x, const int* stop, hipStream_t stream = nullptr)
{
  if (!h || g.size == 1) return;
  if (!stream) stream = g.stream;
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
