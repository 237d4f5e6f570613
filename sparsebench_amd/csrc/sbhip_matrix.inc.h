// sbhip_matrix.inc.h -- part of the single translation unit sbhip.hip (textual include, shares its
// static context): matrices: upload, the compressed mirror (levels 1-5), CRS mirror, accessors.
// ===========================================================================
// matrices
// ===========================================================================
static void* upload(const void* host, size_t bytes)
{
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, bytes ? bytes : 8));
  if (bytes) HIP_CHECK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  return d;
}

static void build_crs_mirror(sb_matrix* m, const uint32_t* rowPtr, const uint32_t* colInd, const double* val);

static void tune_matrix_placement(sb_matrix* m); // (sbhip_launch.inc.h, behind launch_spmv)
static bool g_tunePlacement = true;               // off while an upload builds a device-private mirror (its reference arrays are freed again)

sb_matrix* sb_crs_upload(uint32_t nr, uint32_t nc, const uint32_t* rowPtr, const uint32_t* colInd,
    const double* val)
{
  need_init();
  sb_matrix* m = new sb_matrix();
  m->fmt = 0, m->nr = nr, m->nc = nc, m->nnz = rowPtr[nr];
  for (uint32_t i = 0; i < nr; i++)
    if (rowPtr[i + 1] < rowPtr[i]) SB_FATAL("CRS rowPtr not monotone at row %u", i);
  for (uint32_t k = 0; k < m->nnz; k++)
    if (colInd[k] >= nc) SB_FATAL("CRS colInd[%u]=%u out of range (nc=%u)", k, colInd[k], nc);
  // Row blocks of spmv_crs_stream: as many rows as fit ONE tile of CRS_TILE nonzeros and CRS_THREADS rows; a row
  // longer than the tile gets a block of its own.
  {
    std::vector<uint32_t> rb;
    rb.push_back(0);
    uint32_t r = 0;
    while (r < nr) {
      const uint32_t start = r, base = rowPtr[r];
      while (r < nr && r - start < (uint32_t)CRS_THREADS && rowPtr[r + 1] - base <= (uint32_t)CRS_TILE) r++;
      if (r == start) r++; // single oversize row
      rb.push_back(r);
    }
    m->nRowBlocks = (uint32_t)rb.size() - 1;
    m->rowBlocks  = (uint32_t*)upload(rb.data(), rb.size() * sizeof(uint32_t));
  }
  // Equal nonzero windows of spmv_crs_split (kernels.hip.h) where the longest row allows it
  {
    uint32_t longest = 0;
    for (uint32_t i = 0; i < nr; i++) longest = std::max(longest, rowPtr[i + 1] - rowPtr[i]);
    const char* env = getenv("SB_CRS_KERNEL"); // "stream": keep the row-block kernel (A/B)
    if (longest <= CRS_SPLIT_MAXROW && !(env && strcmp(env, "stream") == 0)) {
      const uint32_t T = ((uint32_t)CRS_TILE + 1u - std::max(longest, 1u)) / 64u * 64u;
      const uint32_t nTiles = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)m->nnz + T - 1) / T));
      std::vector<uint32_t> tr((size_t)nTiles + 1);
      uint32_t r = 0;
      for (uint32_t lb = 0; lb < nTiles; lb++) {
        while (r < nr && (uint64_t)rowPtr[r] < (uint64_t)lb * T) r++;
        tr[lb] = r;
      }
      tr[nTiles]   = nr; // (rows behind the last nonzero -- empty ones -- belong to the last tile)
      m->nCrsTiles = nTiles, m->crsT = T;
      m->tileRow   = (uint32_t*)upload(tr.data(), tr.size() * sizeof(uint32_t));
    }
  }
  m->rowPtr     = (uint32_t*)upload(rowPtr, ((size_t)nr + 1) * sizeof(uint32_t));
  // (64 zeroed elements of slack behind the arrays)
  HIP_CHECK(hipMalloc(&m->colInd, ((size_t)m->nnz + 64) * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&m->val, ((size_t)m->nnz + 64) * sizeof(double)));
  HIP_CHECK(hipMemset(m->colInd + m->nnz, 0, 64 * sizeof(uint32_t)));
  HIP_CHECK(hipMemset(m->val + m->nnz, 0, 64 * sizeof(double)));
  if (m->nnz) {
    HIP_CHECK(hipMemcpy(m->colInd, colInd, (size_t)m->nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->val, val, (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice));
  }
  build_crs_mirror(m, rowPtr, colInd, val);
  tune_matrix_placement(m);
  return m;
}

__global__ void remap_cols_k(uint32_t n, uint32_t nr, const uint32_t* __restrict__ oldToNew,
    uint32_t* colInd)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t c = colInd[i];
    if (c < nr) colInd[i] = oldToNew[c];
  }
}

__global__ void remap_halo_k(uint32_t n, uint32_t nr, const uint32_t* __restrict__ virt, uint32_t* colInd)
{ // halo column nr + h -> nr + virt[h]
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t c = colInd[i];
    if (c >= nr) colInd[i] = nr + virt[c - nr];
  }
}

// Build the compressed mirror of an uploaded SCS C=64 matrix (pack.hip.h).  hostVal is
// the host copy of val (dictionary detection happens on the host, with early exit).
static void build_packed(sb_matrix* m, const double* hostVal, const uint32_t* oldToNewPerm)
{
  if (m->fmt != 1 || m->C != 64 || m->nChunks == 0) return;
  const char* env = getenv("SB_PACK");
  const int want  = env ? atoi(env) : 2; // 0 off, 1 columns only, 2 columns + values
  if (want <= 0) return;
  // 1. value dictionary (<= 256 distinct bit patterns, +0.0 always present for padding)
  std::vector<unsigned long long> dict;
  dict.push_back(0ull);
  bool dictOk = want >= 2;
  if (dictOk) {
    unsigned long long last = 0ull;
    for (size_t i = 0; i < m->nElems; i++) {
      unsigned long long b;
      memcpy(&b, hostVal + i, 8);
      if (b == last) continue;
      last = b;
      if (std::find(dict.begin(), dict.end(), b) == dict.end()) {
        dict.push_back(b);
        if (dict.size() > 256) {
          dictOk = false;
          break;
        }
      }
    }
  }
  std::sort(dict.begin(), dict.end());
  m->nDict  = dictOk ? (int)dict.size() : 0;
  m->padCol = (m->permuted && oldToNewPerm && m->nr) ? oldToNewPerm[0] : 0u;
  // 2. per-chunk column range on the device
  uint32_t *cmin = nullptr, *cmax = nullptr;
  HIP_CHECK(hipMalloc(&cmin, (size_t)m->nChunks * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&cmax, (size_t)m->nChunks * sizeof(uint32_t)));
  const dim3 grid((m->nChunks + 3) / 4), block(256);
  hipLaunchKernelGGL(pack_minmax_k, grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd, m->val,
      m->nChunks, m->padCol, cmin, cmax);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> lo(m->nChunks), hi(m->nChunks), lens(m->nChunks);
  sb_d2h(lo.data(), cmin, lo.size() * sizeof(uint32_t));
  sb_d2h(hi.data(), cmax, hi.size() * sizeof(uint32_t));
  sb_d2h(lens.data(), m->chunkLens, lens.size() * sizeof(uint32_t));
  HIP_CHECK(hipFree(cmin));
  HIP_CHECK(hipFree(cmax));
  // 3. stream positions
  std::vector<PackMeta> meta(m->nChunks);
  uint64_t grp = 0, units = 0;
  m->nWideChunks = 0;
  for (uint32_t c = 0; c < m->nChunks; c++) {
    const bool empty = lo[c] > hi[c];
    const bool wide  = !empty && (hi[c] - lo[c]) >= 0xFFFFu;
    const uint32_t ng = (lens[c] + 3u) / 4u;
    meta[c].grp    = (uint32_t)grp;
    meta[c].idxOff = (uint32_t)units;
    meta[c].base   = empty ? 0u : lo[c];
    meta[c].info   = lens[c] | (wide ? 0x80000000u : 0u);
    grp += ng;
    units += (uint64_t)ng * (wide ? 2u : 1u);
    m->nWideChunks += wide;
  }
  if (grp > 0xFFFFFFFFull || units > 0xFFFFFFFFull) return; // does not fit the 32-bit positions
  m->pmeta = (PackMeta*)upload(meta.data(), meta.size() * sizeof(PackMeta));
  HIP_CHECK(hipMalloc(&m->pidx, (size_t)units * 512 + 1024));
  unsigned long long* dbits = nullptr;
  if (m->nDict) {
    HIP_CHECK(hipMalloc(&m->pcodes, (size_t)grp * 256 + 1024));
    std::vector<unsigned long long> padded(256, 0ull);
    std::copy(dict.begin(), dict.end(), padded.begin());
    dbits    = (unsigned long long*)upload(dict.data(), dict.size() * sizeof(unsigned long long));
    m->pdict = (double*)upload(padded.data(), 256 * sizeof(double));
  }
  hipLaunchKernelGGL(pack_write_k, grid, block, 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd, m->val,
      m->pmeta, dbits, m->nDict, m->nChunks, m->padCol, m->pidx, m->pcodes);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (dbits) HIP_CHECK(hipFree(dbits));
  m->packLevel   = m->nDict ? 2 : 1;
  m->packedBytes = (double)units * 512.0 + (m->nDict ? (double)grp * 256.0 : 8.0 * m->nElems) +
                   16.0 * m->nChunks;
  // default kernel: the packed stream only where it really is smaller.  A matrix whose chunks span more than
  // 65535 columns (far couplings: the irregular stand-in) keeps 32-bit indices and fp64 values -- the same
  // bytes regrouped -- and the reference-layout kernel is the faster one there (260 vs 327 us at 94 M
  // nonzeros).  sb_matrix_use_packed(m, 1) selects it regardless.
  m->usePacked = m->packedBytes <= 0.9 * (12.0 * m->nElems) ? 1 : 0;
}

// The x windows of tiles of `cpt` chunks: per tile the contiguous column ranges its rows touch (columns in the
// DEVICE numbering, i.e. renumbered by the SCS permutation), ranges closer than MERGE_GAP merged, slot 0 reserved
// for x[padCol].  Host arrays are the reference-layout ones (columns in ORIGINAL numbering).  ok = false: some
// tile's window exceeds WMAX entries.
struct TileWindows {
  std::vector<uint32_t> segPtr;
  std::vector<TileSeg> segs;
  uint32_t maxWin = 0;
  uint64_t sumWin = 0, sumElems = 0;
  bool ok = false;
};
static TileWindows compute_tile_windows(const sb_matrix* m, uint32_t cpt, const uint32_t* chunkPtr,
    const uint32_t* chunkLens, const uint32_t* colInd, const double* val, const uint32_t* oldToNewPerm,
    bool original = false, // original: windows in the ORIGINAL column numbering (level 6 of permuted matrices) ...
    const uint32_t* haloVirt = nullptr) // ... with halo column nr + h standing at nr + haloVirt[h]
{
  const uint32_t WMAX = 6144, MERGE_GAP = 8; // window <= 48 KiB of LDS per workgroup
  const uint32_t nTiles = (m->nChunks + cpt - 1) / cpt;
  TileWindows W;
  W.segPtr.assign(nTiles + 1, 0);
  std::vector<TileSeg>& segs = W.segs;
  std::vector<uint32_t> cols;
  std::vector<uint64_t> bitmap;
  for (uint32_t t = 0; t < nTiles; t++) {
    cols.clear();
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint32_t c = t * cpt; c < std::min(t * cpt + cpt, m->nChunks); c++) {
      const size_t cp = chunkPtr[c];
      const size_t n  = (size_t)chunkLens[c] * 64;
      for (size_t e = 0; e < n; e++) {
        uint32_t col = colInd[cp + e];
        unsigned long long bits;
        memcpy(&bits, val + cp + e, 8);
        if (col == 0 && bits == 0) continue; // padding (or an explicit 0.0 at column 0): slot 0
        if (!original && m->permuted && col < m->nr) col = oldToNewPerm[col];
        if (haloVirt && col >= m->nr) col = m->nr + haloVirt[col - m->nr];
        cols.push_back(col);
        lo = std::min(lo, col), hi = std::max(hi, col);
      }
    }
    W.segPtr[t] = (uint32_t)segs.size();
    if (cols.empty()) continue;
    // distinct columns in ascending order: bitmap when the span is modest, sort otherwise
    const uint64_t span = (uint64_t)hi - lo + 1;
    uint32_t win = 1; // slot 0
    auto emit = [&](uint32_t first, uint32_t last) {
      TileSeg sg;
      sg.col = first, sg.len = last - first + 1, sg.lds = win, sg.pad_ = 0;
      win += sg.len;
      segs.push_back(sg);
    };
    if (span <= (1u << 22)) {
      bitmap.assign((span + 63) / 64, 0ull);
      for (uint32_t c : cols) bitmap[(c - lo) >> 6] |= 1ull << ((c - lo) & 63);
      bool open = false;
      uint32_t first = 0, last = 0;
      for (uint64_t w = 0; w < bitmap.size(); w++) {
        uint64_t bits = bitmap[w];
        while (bits) {
          const uint32_t c = lo + (uint32_t)(w * 64 + (uint64_t)__builtin_ctzll(bits));
          bits &= bits - 1;
          if (!open) first = last = c, open = true;
          else if (c - last <= MERGE_GAP) last = c;
          else emit(first, last), first = last = c;
        }
      }
      if (open) emit(first, last);
    } else {
      std::sort(cols.begin(), cols.end());
      uint32_t first = cols[0], last = cols[0];
      for (uint32_t c : cols) {
        if (c - last <= MERGE_GAP) last = std::max(last, c);
        else emit(first, last), first = last = c;
      }
      emit(first, last);
    }
    if (win > WMAX) return W; // some tile's window does not fit LDS
    W.maxWin = std::max(W.maxWin, win);
    W.sumWin += win;
    W.sumElems += cols.size();
  }
  W.segPtr[nTiles] = (uint32_t)segs.size();
  if (W.maxWin == 0) W.maxWin = 1;
  if (segs.empty()) segs.push_back(TileSeg{ 0, 0, 0, 0 });
  W.ok = true;
  return W;
}

// Level 3 of the compressed mirror: per tile (4 chunks = one workgroup) the x window is staged in LDS
// (pack.hip.h: spmv_scs64_lds).
static void build_lds_windows(sb_matrix* m, const uint32_t* chunkPtr, const uint32_t* chunkLens,
    const uint32_t* colInd, const double* val, const uint32_t* oldToNewPerm)
{
  if (m->packLevel < 1) return;
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 3) < 3) return;
  const uint32_t nTiles = (m->nChunks + 3) / 4;
  TileWindows W = compute_tile_windows(m, 4, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
  if (!W.ok) return; // stay at level 1/2
  std::vector<uint32_t>& segPtr = W.segPtr;
  std::vector<TileSeg>& segs    = W.segs;
  const uint32_t maxWin = W.maxWin;
  const uint64_t sumWin = W.sumWin, sumElems = W.sumElems;
  // Staging pays only when a window entry is reused several times and the window is made
  // of long runs (coalesced copies).  Measured: 27-pt stencil reuse 4.6 / run ~510 ->
  // 1.15x faster than gathering through the cache; irregular FE-like matrix with 5 % far
  // couplings reuse 2.5 / run ~3 -> 2.5x slower.  SB_PACK_LDS=1 forces it on, =0 off.
  {
    const double reuse = sumWin ? (double)sumElems / (double)sumWin : 0.0;
    const double run   = segs.empty() ? 0.0 : (double)sumWin / (double)segs.size();
    const char* force  = getenv("SB_PACK_LDS");
    // (reuse: 2.75, not 3 -- a 27-point stencil whose x-lines are exactly one 4-chunk tile long, n = 256, sits at 2.98 with these
    //  tiles and at 4.5 with the 8-chunk tiles of level 6, which this decision gates; the run length keeps irregular matrices out)
    const bool want    = force ? atoi(force) != 0 : (reuse >= 2.75 && run >= 32.0);
    if (!want) return;
  }
  m->tileSegPtr = (uint32_t*)upload(segPtr.data(), segPtr.size() * sizeof(uint32_t));
  m->tileSegs   = (TileSeg*)upload(segs.data(), segs.size() * sizeof(TileSeg));
  std::vector<PackMeta> meta(m->nChunks);
  sb_d2h(meta.data(), m->pmeta, meta.size() * sizeof(PackMeta));
  const uint64_t groups = meta.empty() ? 0 : (uint64_t)meta.back().grp + ((meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
  HIP_CHECK(hipMalloc(&m->pslots, (size_t)groups * 512 + 1024));
  hipLaunchKernelGGL(pack_slots_k, dim3(nTiles), dim3(256), 0, g.stream, m->chunkPtr, m->chunkLens, m->colInd,
      m->val, m->pmeta, m->tileSegPtr, m->tileSegs, m->nChunks, m->padCol, m->pslots, 4u);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  m->ldsWindow = maxWin;
  m->slotBytes = (double)groups * 512.0 + (m->nDict ? (double)groups * 256.0 : 8.0 * m->nElems) +
                 16.0 * m->nChunks + 16.0 * segs.size() + 4.0 * nTiles;
  m->usePacked = 2; // 2: packed stream + x window in LDS
}

// ---- levels 4 and 5 (pack.hip.h): host side ---------------------------------------------------
// Level 4: one byte per element naming a (value, slot delta) pair of the tile's class;
// level 5: per chunk one shared row pattern + the odd lanes.  Needs the value dictionary and the
// LDS windows.  SB_PACK=4 stops at level 4 (every chunk per-lane).
struct PatternPlan { // working set of build_patterns
  uint32_t nTiles = 0;
  uint32_t cpt    = 4;           // chunks per tile of the pattern kernel (4: level 3's tiles; 8: own windows)
  const uint32_t* slots = nullptr; // device: 16-bit window slots of every element for THIS tiling
  std::vector<uint32_t> segPtr;  // the tiling's windows (host copies)
  std::vector<TileSeg> segs;
  // level 4
  std::vector<std::vector<uint32_t>> classes; // sorted pair keys of every class
  std::vector<uint32_t> tileClass;
  std::vector<PatEntry> classDict; // [class][256]
  std::vector<PackMeta> meta;
  uint64_t groups = 0; // code groups (4 columns) of the whole matrix
  // level 5
  std::vector<uint32_t> exc;      // [chunk][2] exception-lane mask
  std::vector<PatEntry> rowPats;  // shared dominant row patterns, back to back
  size_t nRowPats = 0;
  std::vector<uint32_t> chunkOff, chunkFlags, chunkPat, tileExcStart, tileExcCount;
  uint64_t words = 0, excEntries = 0; // L code words / U exception entries in total
  uint32_t excLds = 0;                // most exception entries of one tile
  bool anyL       = false;
  // device side of the analysis (owned by whoever keeps the form)
  uint16_t* dRowBase   = nullptr; // per row: window slot of its first element
  uint32_t* dTileClass = nullptr;
  PatEntry* dClassDict = nullptr;
  uint32_t* dLanes     = nullptr; // per-lane code words of every chunk (temporary)
  uint32_t* dExc       = nullptr; // [chunk][2] exception-lane mask (temporary)
  std::vector<uint32_t> dom;      // dominant code words per group
  // level 6 with the window in ORIGINAL column order (sigma > 1): slot -> device column
  bool mapped = false;
  uint32_t mapStride = 0;              // 16-bit map entries per tile (a multiple of 256)
  std::vector<uint16_t> slotMap;       // [tile][mapStride]: device column - blockBase[slot / 256]
  std::vector<uint32_t> blockBase;     // [tile][18]
  void free_temporaries() { sb_free(dLanes), sb_free(dExc), dLanes = nullptr, dExc = nullptr; }
  void free_tables() { sb_free(dRowBase), sb_free(dTileClass), sb_free(dClassDict), dRowBase = nullptr, dTileClass = nullptr, dClassDict = nullptr; }
};

// pairs of every tile (device) -> classes of <= PAT_MAX pairs -> class tables; false: level 3 stays
static bool pattern_classes(sb_matrix* m, PatternPlan& P)
{
  const uint32_t nTiles = P.nTiles;
  uint32_t *dCount = nullptr, *dKeys = nullptr;
  HIP_CHECK(hipMalloc(&P.dRowBase, (size_t)m->nChunks * 64 * sizeof(uint16_t) + 16));
  HIP_CHECK(hipMalloc(&dCount, (size_t)nTiles * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&dKeys, (size_t)nTiles * 256 * sizeof(uint32_t)));
  hipLaunchKernelGGL(pat_collect_k, dim3(nTiles), dim3(64 * P.cpt), 0, g.stream, m->pmeta, P.slots, m->pcodes,
      m->nChunks, P.dRowBase, dCount, dKeys);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> count(nTiles), keys((size_t)nTiles * 256);
  sb_d2h(count.data(), dCount, count.size() * sizeof(uint32_t));
  sb_d2h(keys.data(), dKeys, keys.size() * sizeof(uint32_t));
  sb_free(dCount), sb_free(dKeys);
  for (uint32_t t = 0; t < nTiles; t++)
    if (count[t] > PAT_MAX) {
      if (getenv("SB_PACK_REPORT"))
        fprintf(stderr, "sbhip pack: tiles of %u chunks: tile %u holds more than %u (value, slot delta) pairs\n", P.cpt, t, PAT_MAX);
      if (getenv("SB_PACK_DEBUG")) {
        fprintf(stderr, "  mapped %d; segments of tile %u:", (int)P.mapped, t);
        for (uint32_t q = P.segPtr[t]; q < P.segPtr[t + 1]; q++) fprintf(stderr, " [col %u len %u slot %u]", P.segs[q].col, P.segs[q].len, P.segs[q].lds);
        fprintf(stderr, "\n");
      }
      return false;
    }
  // tiles -> classes: a class that already holds the tile's pairs, else the first class the
  // pairs still fit into, else a new class
  P.tileClass.assign(nTiles, 0);
  std::vector<uint32_t> merged;
  const size_t maxClasses = std::max<size_t>(64, nTiles / 4);
  uint32_t lastClass = 0;
  auto& classes = P.classes;
  for (uint32_t t = 0; t < nTiles; t++) {
    uint32_t* k = keys.data() + (size_t)t * 256;
    std::sort(k, k + count[t]);
    int found = -1;
    if (!classes.empty() && std::includes(classes[lastClass].begin(), classes[lastClass].end(), k, k + count[t]))
      found = (int)lastClass;
    for (size_t c = 0; found < 0 && c < classes.size(); c++)
      if (std::includes(classes[c].begin(), classes[c].end(), k, k + count[t])) found = (int)c;
    for (size_t c = 0; found < 0 && c < classes.size(); c++) {
      merged.clear();
      std::set_union(classes[c].begin(), classes[c].end(), k, k + count[t], std::back_inserter(merged));
      if (merged.size() <= PAT_MAX) classes[c] = merged, found = (int)c;
    }
    if (found < 0) {
      if (classes.size() >= maxClasses) { // no repeating patterns: not worth the tables
        if (getenv("SB_PACK_REPORT")) fprintf(stderr, "sbhip pack: tiles of %u chunks: more than %zu pattern classes\n", P.cpt, maxClasses);
        return false;
      }
      classes.emplace_back(k, k + count[t]);
      found = (int)classes.size() - 1;
    }
    P.tileClass[t] = lastClass = (uint32_t)found;
  }
  if (classes.empty()) classes.emplace_back();
  P.meta.resize(m->nChunks);
  sb_d2h(P.meta.data(), m->pmeta, P.meta.size() * sizeof(PackMeta));
  for (const PackMeta& pm : P.meta)
    if ((pm.info & 0x7FFFFFFFu) > PAT_LEN_MASK) return false; // chunk width collides with the header's flag bits
  P.groups = (uint64_t)P.meta.back().grp + ((P.meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
  std::vector<double> dict(256);
  sb_d2h(dict.data(), m->pdict, 256 * sizeof(double));
  P.classDict.assign(classes.size() * 256, PatEntry{ 0.0, 0, 0u });
  for (size_t c = 0; c < classes.size(); c++)
    for (size_t i = 0; i < classes[c].size(); i++) {
      const uint32_t key = classes[c][i];
      PatEntry& e        = P.classDict[c * 256 + i];
      e.v                = dict[key & 255u];
      if (key & PAT_ABS) e.off8 = 0u, e.m = 0u; // padding: slot 0
      else e.off8 = (uint32_t)(8 * ((int32_t)((key >> 8) & 0xFFFFu) - 32768)), e.m = 1u; // 8 * (slot - rowBase), mod 2^32
    }
  return true;
}

// class tables to the device; per-lane code words of every chunk (its L form); the dominant sequence of every
// chunk + its odd lanes
static void pattern_codes(sb_matrix* m, PatternPlan& P)
{
  std::vector<uint32_t> classKeys(P.classes.size() * 256, PAT_EMPTY);
  for (size_t c = 0; c < P.classes.size(); c++) std::copy(P.classes[c].begin(), P.classes[c].end(), classKeys.begin() + c * 256);
  uint32_t* dClassKeys = (uint32_t*)upload(classKeys.data(), classKeys.size() * sizeof(uint32_t));
  P.dTileClass         = (uint32_t*)upload(P.tileClass.data(), P.tileClass.size() * sizeof(uint32_t));
  P.dClassDict         = (PatEntry*)upload(P.classDict.data(), P.classDict.size() * sizeof(PatEntry));
  uint32_t* dDom = nullptr;
  HIP_CHECK(hipMalloc(&P.dLanes, (size_t)P.groups * 256 + 1024));
  HIP_CHECK(hipMalloc(&dDom, (size_t)P.groups * sizeof(uint32_t) + 16));
  HIP_CHECK(hipMalloc(&P.dExc, (size_t)m->nChunks * 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(pat_encode_k, dim3(P.nTiles), dim3(64 * P.cpt), 0, g.stream, m->pmeta, P.slots, m->pcodes,
      m->nChunks, P.dRowBase, P.dTileClass, dClassKeys, P.dLanes);
  hipLaunchKernelGGL(pat_dominant_k, dim3((m->nChunks + 3) / 4), dim3(256), 0, g.stream, m->pmeta, P.dLanes, m->nChunks, dDom,
      P.dExc);
  HIP_CHECK(hipGetLastError());
  P.dom.resize(P.groups ? P.groups : 1);
  P.exc.resize((size_t)m->nChunks * 2);
  sb_d2h(P.dom.data(), dDom, (size_t)P.groups * sizeof(uint32_t));
  sb_d2h(P.exc.data(), P.dExc, P.exc.size() * sizeof(uint32_t));
  sb_free(dDom), sb_free(dClassKeys);
}

// chunk by chunk: U (row pattern + expanded exception lanes) or L (code words of all 64 lanes); row
// patterns are shared between chunks (key: the expanded entries).  A tile's exception entries are
// staged in LDS, so a tile with too many of them stays L.  `dom`: dominant code words per group.
static void pattern_rows(sb_matrix* m, PatternPlan& P, const std::vector<uint32_t>& dom, bool wantRows)
{
  const size_t maxPatEntries = 1u << 20; // 16 MiB of pattern rows at most
  std::unordered_map<std::string, uint32_t> patIndex;
  P.chunkOff.assign(m->nChunks, 0), P.chunkFlags.assign(m->nChunks, 0), P.chunkPat.assign(m->nChunks, 0);
  P.tileExcStart.assign(P.nTiles, 0), P.tileExcCount.assign(P.nTiles, 0);
  std::vector<PatEntry> row;
  m->nUniformChunks = 0;
  auto n_exc = [&](uint32_t c) {
    return (uint32_t)__builtin_popcount(P.exc[2 * (size_t)c]) + (uint32_t)__builtin_popcount(P.exc[2 * (size_t)c + 1]);
  };
  for (uint32_t t = 0; t < P.nTiles; t++) {
    const uint32_t c0 = t * P.cpt, c1 = std::min(c0 + P.cpt, m->nChunks);
    uint64_t tileExc = 0;
    bool tileOk      = wantRows;
    for (uint32_t c = c0; c < c1 && tileOk; c++) {
      const uint32_t len = P.meta[c].info & 0x7FFFFFFFu;
      if (len == 0 || n_exc(c) > PAT_EXC_MAX) continue; // this chunk will be L
      tileExc += (uint64_t)n_exc(c) * len;
    }
    if (tileExc > PAT_EXC_LDS_MAX * (P.cpt / 4)) tileOk = false;
    P.tileExcStart[t] = (uint32_t)P.excEntries;
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t len = P.meta[c].info & 0x7FFFFFFFu, ng = (len + 3u) / 4u, nExc = n_exc(c);
      bool uni = tileOk && len > 0 && nExc <= PAT_EXC_MAX, nopad = true;
      if (uni) {
        row.resize(len);
        const PatEntry* cd = P.classDict.data() + (size_t)P.tileClass[t] * 256;
        for (uint32_t j = 0; j < len; j++) {
          row[j] = cd[(dom[P.meta[c].grp + j / 4] >> (8u * (j & 3u))) & 255u];
          nopad  = nopad && row[j].m == 1u;
        }
        std::string key((const char*)row.data(), row.size() * sizeof(PatEntry));
        auto it = patIndex.find(key);
        if (it != patIndex.end()) P.chunkPat[c] = it->second;
        else if (P.rowPats.size() + len <= maxPatEntries) {
          P.chunkPat[c] = (uint32_t)P.rowPats.size();
          patIndex.emplace(std::move(key), P.chunkPat[c]);
          P.rowPats.insert(P.rowPats.end(), row.begin(), row.end());
        } else uni = false; // table full
      }
      if (uni) {
        P.chunkOff[c]   = (uint32_t)P.excEntries;
        P.chunkFlags[c] = len | PAT_UNIFORM | (nopad ? PAT_NOPAD : 0u);
        P.excEntries += (uint64_t)nExc * len;
        P.tileExcCount[t] += nExc * len;
        m->nUniformChunks++;
      } else {
        P.chunkOff[c]   = (uint32_t)P.words;
        P.chunkFlags[c] = len;
        P.words += (uint64_t)ng * 64u;
        P.anyL = P.anyL || len > 0;
      }
    }
    P.excLds = std::max(P.excLds, P.tileExcCount[t]);
  }
  P.nRowPats = patIndex.size();
  if (P.rowPats.empty()) P.rowPats.push_back(PatEntry{ 0.0, 0u, 0u });
}

// one header per tile: class, chunk positions / widths / row patterns, the first segments; tiles
// whose window holds a halo column (>= nr) go last, so that the interior part of the product does
// not have to wait for the halo exchange (loop_body).  4-chunk tiles: one TileHdr (48 words); 8-chunk tiles: two
// TileHdr halves interleaved word by word into 128 words (X: the tile-level fields + chunks 0-3, Y: the per-chunk
// fields of chunks 4-7), which the kernel fetches as one 8-byte vector load.  Returns the number of segments.
// tileOfHdr (optional): which tile the i-th stored header describes
static size_t pattern_headers(sb_matrix* m, const PatternPlan& P, uint32_t** out, uint32_t* interiorOut,
    std::vector<uint32_t>* tileOfHdr = nullptr, bool* allSimpleOut = nullptr)
{
  const uint32_t nTiles = P.nTiles, cpt = P.cpt, LONG = cpt == 8 ? 4u : 3u;
  const std::vector<uint32_t>& segPtr = P.segPtr;
  const std::vector<TileSeg>& segs    = P.segs;
  const size_t nSegs = segPtr[nTiles];
  struct Pair { TileHdr x, y; };
  std::vector<Pair> hdrs(nTiles);
  for (uint32_t t = 0; t < nTiles; t++) {
    TileHdr& h = hdrs[t].x;
    memset(&hdrs[t], 0, sizeof(Pair));
    h.tile = t;
    h.cls = P.tileClass[t], h.nseg = segPtr[t + 1] - segPtr[t], h.segPtr = segPtr[t], h.win = 1;
    h.excStart = P.tileExcStart[t], h.excCount = P.tileExcCount[t];
    for (uint32_t w = 0; w < cpt; w++) {
      const uint32_t c = t * cpt + w;
      if (c >= m->nChunks) continue;
      TileHdr& hh = w < 4 ? hdrs[t].x : hdrs[t].y;
      const uint32_t k = w & 3u;
      hh.off[k] = P.chunkOff[c], hh.len[k] = P.chunkFlags[c], hh.rowPat[k] = P.chunkPat[c];
      if (P.chunkFlags[c] & PAT_UNIFORM) hh.exc[k][0] = P.exc[2 * (size_t)c], hh.exc[k][1] = P.exc[2 * (size_t)c + 1];
    }
    for (uint32_t s2 = 0; s2 < PAT_INLINE_SEGS; s2++) h.seg[s2][1] = 0xFFFFFFFFu;
    h.winInline = 1;
    // simple window: <= 6 segments which, longest first, are 3 x <= 256 * LONG and 3 x <= 256 entries
    std::vector<TileSeg> ts(segs.begin() + segPtr[t], segs.begin() + segPtr[t] + h.nseg);
    std::stable_sort(ts.begin(), ts.end(), [](const TileSeg& a, const TileSeg& b) { return a.len > b.len; });
    bool simple = h.nseg <= PAT_INLINE_SEGS;
    for (uint32_t s2 = 0; s2 < h.nseg && simple; s2++) simple = ts[s2].len <= (s2 < 3 ? 256u * LONG : 256u);
    for (uint32_t s2 = 0; s2 < h.nseg; s2++) {
      const TileSeg& sg = simple ? ts[s2] : segs[segPtr[t] + s2]; // slot order unless simple
      if (s2 < PAT_INLINE_SEGS) h.seg[s2][0] = sg.col, h.seg[s2][1] = sg.lds, h.seg[s2][2] = sg.len;
    }
    for (uint32_t s2 = 0; s2 < h.nseg; s2++) {
      const TileSeg& sg = segs[segPtr[t] + s2];
      if (s2 < PAT_INLINE_SEGS) h.winInline = sg.lds + sg.len;
      h.win = sg.lds + sg.len;
    }
    h.flags = simple ? PAT_SIMPLE_WINDOW : 0u;
    if (P.mapped) { // staged through the slot map: the segment words hold the base column of every 256 slots
      h.flags = PAT_MAPPED_WINDOW;
      memcpy(&h.seg[0][0], P.blockBase.data() + (size_t)t * 18, 18 * sizeof(uint32_t));
    }
  }
  *interiorOut = nTiles;
  if (m->nc > m->nr) {
    auto touches_halo = [&](const TileHdr& h) {
      for (uint32_t s2 = 0; s2 < h.nseg; s2++) {
        const TileSeg& sg = segs[h.segPtr + s2];
        if (sg.col + sg.len > m->nr) return true;
      }
      return false;
    };
    for (Pair& h : hdrs)
      if (touches_halo(h.x)) h.x.flags |= PAT_TOUCHES_HALO;
    auto mid = std::stable_partition(hdrs.begin(), hdrs.end(), [&](const Pair& h) { return !(h.x.flags & PAT_TOUCHES_HALO); });
    *interiorOut = (uint32_t)(mid - hdrs.begin());
  }
  if (allSimpleOut) {
    *allSimpleOut = true;
    for (uint32_t t = 0; t < nTiles; t++)
      if (!(hdrs[t].x.flags & PAT_SIMPLE_WINDOW)) *allSimpleOut = false;
  }
  if (tileOfHdr) {
    tileOfHdr->resize(nTiles);
    for (uint32_t t = 0; t < nTiles; t++) (*tileOfHdr)[t] = hdrs[t].x.tile;
  }
  const uint32_t stride = cpt == 8 ? 128u : 48u;
  std::vector<uint32_t> words((size_t)nTiles * stride + 128, 0u);
  for (uint32_t t = 0; t < nTiles; t++) {
    const uint32_t* wx = reinterpret_cast<const uint32_t*>(&hdrs[t].x);
    const uint32_t* wy = reinterpret_cast<const uint32_t*>(&hdrs[t].y);
    uint32_t* o        = words.data() + (size_t)t * stride;
    if (cpt == 8)
      for (uint32_t i = 0; i < 48; i++) o[2 * i] = wx[i], o[2 * i + 1] = wy[i];
    else memcpy(o, wx, 48 * sizeof(uint32_t));
  }
  *out = (uint32_t*)upload(words.data(), words.size() * sizeof(uint32_t));
  if (getenv("SB_PACK_REPORT") && out == &m->tileHdrs) {
    size_t nSimple = 0;
    for (const Pair& h : hdrs) nSimple += h.x.flags & PAT_SIMPLE_WINDOW;
    fprintf(stderr, "sbhip pack: %u tiles of %u chunks (%u interior, %zu simple windows, max %u entries), %zu classes, %u/%u U chunks, "
                    "%zu row patterns (%zu entries), %llu exception entries (max %u per tile), %llu code words\n",
        nTiles, cpt, *interiorOut, nSimple, m->patWindow, P.classes.size(), m->nUniformChunks, m->nChunks, P.nRowPats,
        P.rowPats.size(), (unsigned long long)P.excEntries, P.excLds, (unsigned long long)P.words);
  }
  return nSegs;
}

// Level 6 (pack.hip.h): every chunk as a row program of its tile + per-row base slots, where the rows allow it.
// `dom`: dominant code words per group; `lanes` (device): per-lane code words of every chunk.  Built next to the
// level-5 form (same tiles, windows, classes); kept when nearly all chunks qualify.
// P: a finished analysis (windows, classes, per-lane codes, dominant sequences).  true: kept (m->m* set).
static bool build_masked(sb_matrix* m, const PatternPlan& P)
{
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 6) < 6 || m->nChunks == 0) return false;
  const std::vector<uint32_t>& dom = P.dom;
  std::vector<uint32_t> codes((size_t)P.groups * 64);
  std::vector<uint16_t> rowBase((size_t)m->nChunks * 64);
  sb_d2h(codes.data(), P.dLanes, codes.size() * sizeof(uint32_t));
  sb_d2h(rowBase.data(), P.dRowBase, rowBase.size() * sizeof(uint16_t));
  struct Ent { uint64_t v; int32_t off; }; // value bits, byte offset relative to 8 * base slot
  PatternPlan Q;
  Q.nTiles = P.nTiles, Q.cpt = P.cpt, Q.segPtr = P.segPtr, Q.segs = P.segs, Q.tileClass = P.tileClass;
  Q.mapped = P.mapped, Q.mapStride = P.mapStride, Q.blockBase = P.blockBase;
  Q.chunkOff.assign(m->nChunks, 0), Q.chunkFlags.assign(m->nChunks, 0), Q.chunkPat.assign(m->nChunks, 0);
  Q.tileExcStart.assign(P.nTiles, 0), Q.tileExcCount.assign(P.nTiles, 0);
  Q.exc.assign((size_t)m->nChunks * 2, 0);
  std::vector<int16_t> mBase((size_t)m->nChunks * 64, 0);
  for (size_t i = 0; i < mBase.size(); i++) mBase[i] = (int16_t)rowBase[i]; // (L chunks keep the slot of the row's first element)
  std::vector<ProgBlock> progs;
  std::unordered_map<std::string, uint32_t> progIndex;
  const size_t maxProgBlocks = 1u << 16; // 12 MiB of programs at most
  uint32_t nMasked = 0, nNonEmpty = 0;
  std::vector<std::vector<Ent>> cand; // the tile's candidate programs: dominant rows of its chunks, longest first
  std::vector<Ent> rowE;
  std::vector<ProgBlock> prog;
  std::vector<uint64_t> emask;
  std::vector<int32_t> base(64);
  for (uint32_t t = 0; t < P.nTiles; t++) {
    const uint32_t c0 = t * P.cpt, c1 = std::min(c0 + P.cpt, m->nChunks);
    const PatEntry* cd = P.classDict.data() + (size_t)P.tileClass[t] * 256;
    // entries of a code sequence (real ones; false: a padding entry in front of a real one -- not handled here)
    auto expand = [&](auto&& code_at, uint32_t len, std::vector<Ent>& out) {
      out.clear();
      bool padSeen = false;
      for (uint32_t j = 0; j < len; j++) {
        const PatEntry& e = cd[code_at(j)];
        if (e.m == 0u) { padSeen = true; continue; }
        if (padSeen) return false;
        uint64_t vb;
        memcpy(&vb, &e.v, 8);
        out.push_back(Ent{ vb, (int32_t)e.off8 });
      }
      return true;
    };
    cand.clear();
    std::vector<int> candOf(c1 - c0, -1);
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t len = P.meta[c].info & 0x7FFFFFFFu;
      if (!len) continue;
      std::vector<Ent> d;
      const uint32_t grp = P.meta[c].grp;
      if (!expand([&](uint32_t j) { return (dom[grp + j / 4] >> (8u * (j & 3u))) & 255u; }, len, d) || d.empty()) continue;
      int found = -1;
      for (size_t k = 0; k < cand.size() && found < 0; k++)
        if (cand[k].size() == d.size() && !memcmp(cand[k].data(), d.data(), d.size() * sizeof(Ent))) found = (int)k;
      if (found < 0) cand.push_back(std::move(d)), found = (int)cand.size() - 1;
      candOf[c - c0] = found;
    }
    std::vector<int> order(cand.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cand[a].size() > cand[b].size(); });
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t len = P.meta[c].info & 0x7FFFFFFFu, ng = (len + 3u) / 4u, grp = P.meta[c].grp;
      if (!len) continue;
      nNonEmpty++;
      bool done = false;
      // its own dominant row first, then the tile's other rows from the longest down
      for (int attempt = -1; attempt < (int)order.size() && !done; attempt++) {
        const int k = attempt < 0 ? candOf[c - c0] : order[attempt];
        if (k < 0 || (attempt >= 0 && k == candOf[c - c0])) continue;
        const std::vector<Ent>& D = cand[k];
        const uint32_t dl = (uint32_t)D.size();
        if (dl > 64u * 1024u) continue;
        emask.assign(dl, 0ull);
        uint64_t padMask = 0;
        bool ok          = true;
        for (uint32_t lane = 0; lane < 64 && ok; lane++) {
          const uint32_t* cw = codes.data() + (size_t)grp * 64 + lane;
          if (!expand([&](uint32_t j) { return (cw[(size_t)(j / 4) * 64] >> (8u * (j & 3u))) & 255u; }, len, rowE)) { ok = false; break; }
          const uint32_t r   = (uint32_t)rowE.size();
          const int32_t rb8  = 8 * (int32_t)rowBase[(size_t)c * 64 + lane];
          if (r < len) padMask |= 1ull << lane; // the row has reference padding
          base[lane] = 0;
          if (r == 0) continue;
          if (r > dl) { ok = false; break; }
          bool fit = false;
          for (uint32_t j0 = 0; j0 + r <= dl && !fit; j0++) {
            if (D[j0].v != rowE[0].v) continue;
            const int32_t b8 = rb8 + rowE[0].off - D[j0].off; // base that puts the row's first entry on program entry j0
            if (b8 % 8 != 0 || b8 / 8 < -32768 || b8 / 8 > 32767) continue;
            uint32_t kk = 0; // greedy, left to right: the row's entries as a sub-sequence of the program
            for (uint32_t j = j0; j < dl && kk < r; j++)
              if (D[j].v == rowE[kk].v && b8 + D[j].off == rb8 + rowE[kk].off) kk++;
            if (kk != r) continue;
            kk = 0;
            for (uint32_t j = j0; j < dl && kk < r; j++)
              if (D[j].v == rowE[kk].v && b8 + D[j].off == rb8 + rowE[kk].off) emask[j] |= 1ull << lane, kk++;
            base[lane] = b8 / 8;
            fit        = true;
          }
          if (!fit) ok = false;
        }
        if (!ok) continue;
        bool full = true;
        prog.assign((dl + 7) / 8, ProgBlock{});
        for (uint32_t j = 0; j < dl; j++) {
          ProgBlock& pb = prog[j / 8];
          memcpy(&pb.v[j & 7u], &D[j].v, 8), pb.mask[j & 7u] = emask[j], pb.off8[j & 7u] = (uint32_t)D[j].off;
          full = full && emask[j] == ~0ull;
        }
        std::string key((const char*)prog.data(), prog.size() * sizeof(ProgBlock));
        auto it = progIndex.find(key);
        uint32_t at;
        if (it != progIndex.end()) at = it->second;
        else if (progs.size() + prog.size() <= maxProgBlocks) {
          at = (uint32_t)progs.size();
          progIndex.emplace(std::move(key), at);
          progs.insert(progs.end(), prog.begin(), prog.end());
        } else break; // table full: L chunk
        Q.chunkPat[c]   = at;
        Q.chunkFlags[c] = dl | PAT_UNIFORM | (full ? PAT_NOPAD : 0u) | (padMask ? PAT_HASPAD : 0u);
        Q.exc[2 * (size_t)c] = (uint32_t)padMask, Q.exc[2 * (size_t)c + 1] = (uint32_t)(padMask >> 32);
        for (uint32_t lane = 0; lane < 64; lane++) mBase[(size_t)c * 64 + lane] = (int16_t)base[lane];
        nMasked++;
        done = true;
      }
      if (!done) {
        Q.chunkOff[c]   = (uint32_t)Q.words;
        Q.chunkFlags[c] = len;
        Q.words += (uint64_t)ng * 64u;
        Q.anyL = true;
      }
    }
  }
  const bool keep = nNonEmpty > 0 && (double)nMasked >= 0.98 * (double)nNonEmpty && Q.words <= 0xFFFFFFFFull;
  if (getenv("SB_PACK_REPORT"))
    fprintf(stderr, "sbhip pack: masked form%s: %u/%u chunks as row programs (%zu programs, %zu blocks of 8 entries), %llu code words%s\n",
        P.mapped ? " (windows in original column order)" : "", nMasked, nNonEmpty, progIndex.size(), progs.size(),
        (unsigned long long)Q.words, keep ? "" : " -- not kept");
  if (!keep) return false;
  progs.push_back(ProgBlock{}); // (never empty)
  uint32_t* dOff   = (uint32_t*)upload(Q.chunkOff.data(), Q.chunkOff.size() * sizeof(uint32_t));
  uint32_t* dFlags = (uint32_t*)upload(Q.chunkFlags.data(), Q.chunkFlags.size() * sizeof(uint32_t));
  const size_t streamBytes = (size_t)Q.words * sizeof(uint32_t) + 1024;
  HIP_CHECK(hipMalloc(&m->mStream, streamBytes));
  HIP_CHECK(hipMemsetAsync(m->mStream, 0, streamBytes, g.stream));
  hipLaunchKernelGGL(pat_compact_k, dim3((m->nChunks + 3) / 4), dim3(256), 0, g.stream, m->pmeta, P.dLanes, m->nChunks, dOff,
      dFlags, P.dExc, P.dRowBase, P.dTileClass, P.dClassDict, m->mStream, (PatEntry*)nullptr, P.cpt);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  sb_free(dOff), sb_free(dFlags);
  HIP_CHECK(hipMalloc(&m->mRowBase, mBase.size() * sizeof(int16_t) + 16));
  HIP_CHECK(hipMemcpy(m->mRowBase, mBase.data(), mBase.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  m->mProgs        = (ProgBlock*)upload(progs.data(), progs.size() * sizeof(ProgBlock));
  m->nProgs        = (uint32_t)progIndex.size();
  m->nMaskedChunks = nMasked;
  m->mDict         = Q.anyL ? 256u : 0u;
  m->mClassDict    = P.dClassDict; // (shared with the level-5 form unless the windows are the mapped ones)
  const size_t nSegs = pattern_headers(m, Q, &m->mHdrs, &m->mInterior, &m->mTileOfHdr, &m->mAllSimple);
  m->mCPT = P.cpt, m->mNTiles = P.nTiles;
  m->mBytes = (double)Q.words * 4.0 + 2.0 * 64.0 * m->nChunks + (P.mapped ? 2.0 * P.mapStride * P.nTiles : 16.0 * nSegs) +
              (double)(P.cpt == 8 ? 384 : 192) * P.nTiles + (Q.anyL ? 4096.0 * P.classes.size() : 0.0) + 192.0 * progs.size();
  return true;
}

// Level 6 for a row-permuted matrix (sigma > 1).  In the device numbering the sort has moved the short rows of
// every sigma window to its end, so the neighbours of the rows next to them sit at odd distances: no shared
// program.  The window is therefore laid out in ORIGINAL column order -- there a row and its neighbours keep
// their distances whatever the sort did -- and staged through a 16-bit map, slot -> device column (relative to
// a base per 256 slots), instead of segment by segment.
static bool build_masked_mapped(sb_matrix* m, uint32_t cpt, const uint32_t* chunkPtr, const uint32_t* chunkLens,
    const uint32_t* colInd, const double* val)
{
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 6) < 6 || m->nChunks == 0) return false;
  // Halo columns (>= nr) are numbered by the partitioner in the order it met them (src/comm.c:60-110: first seen inside
  // an owner's group): next to a rank boundary that interleaves the neighbour's first grid lines, and one tile with > 255
  // (value, slot delta) pairs costs the whole matrix its pattern levels.  The window is private: where the host has
  // passed the halo columns' global ids (sb_set_external_ids, from commPartition), they stand in it in ascending global
  // order -- for a stencil the neighbour's plane, line by line, like the rank's own planes.
  const uint32_t nExt = m->nc - m->nr;
  std::vector<uint32_t> haloVirt, haloReal;
  if (nExt && g_externalIds.size() == nExt) {
    haloReal.resize(nExt), haloVirt.resize(nExt);
    for (uint32_t h = 0; h < nExt; h++) haloReal[h] = h;
    std::stable_sort(haloReal.begin(), haloReal.end(), [&](uint32_t a2, uint32_t b2) { return g_externalIds[a2] < g_externalIds[b2]; });
    for (uint32_t v = 0; v < nExt; v++) haloVirt[haloReal[v]] = v;
  }
  const bool virt = !haloVirt.empty();
  TileWindows W = compute_tile_windows(m, cpt, chunkPtr, chunkLens, colInd, val, nullptr, true, virt ? haloVirt.data() : nullptr);
  if (!W.ok) return false;
  PatternPlan P;
  P.cpt = cpt, P.nTiles = (m->nChunks + cpt - 1) / cpt, P.mapped = true;
  // slot -> device column; a block of 256 slots must span < 65536 device columns: where it would not, the
  // segment that starts inside the block is moved up to the next block boundary
  std::vector<uint32_t> o2n;
  if (m->permuted) {
    o2n.resize(m->nr);
    sb_d2h(o2n.data(), m->oldToNew, o2n.size() * sizeof(uint32_t));
  }
  auto dev_col = [&](uint32_t orig) { // window (original / virtual halo) numbering -> device column
    if (orig >= m->nr) return virt ? m->nr + haloReal[orig - m->nr] : orig;
    return m->permuted ? o2n[orig] : orig;
  };
  uint32_t maxWin = 1;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      P.mapStride = ((maxWin + 255u) / 256u) * 256u;
      if (P.mapStride > (cpt == 8 ? 15u : 12u) * 256u) return false; // what a workgroup stages in one pass (spmv_scs64_pat: WB); block bases: <= 18 header words
      P.slotMap.assign((size_t)P.nTiles * P.mapStride, 0), P.blockBase.assign((size_t)P.nTiles * 18, 0);
    }
    for (uint32_t t = 0; t < P.nTiles; t++) {
      const uint32_t s0 = W.segPtr[t], s1 = W.segPtr[t + 1];
      if (pass == 0) { // slots: keep a block's device columns within 16 bits
        uint32_t win = 1, blk = 0xFFFFFFFFu, lo = 0, hi = 0;
        for (uint32_t s2 = s0; s2 < s1; s2++) {
          TileSeg& sg = W.segs[s2];
          for (int tries = 0; tries < 2; tries++) {
            uint32_t w = win, b = blk, l = lo, h = hi;
            bool fits = true;
            for (uint32_t i = 0; i < sg.len && fits; i++, w++) {
              const uint32_t d = dev_col(sg.col + i);
              if (w / 256u != b) b = w / 256u, l = h = d;
              l = std::min(l, d), h = std::max(h, d);
              fits = h - l < 65536u;
            }
            if (fits) { sg.lds = win, win = w, blk = b, lo = l, hi = h; break; }
            if (tries == 1) return false; // a single segment whose device columns are too far apart
            win = (win + 255u) / 256u * 256u; // start the segment on a block boundary
          }
        }
        if (win > 6144u) return false;
        maxWin = std::max(maxWin, win);
      } else {
        uint16_t* mp  = P.slotMap.data() + (size_t)t * P.mapStride;
        uint32_t* bb  = P.blockBase.data() + (size_t)t * 18;
        std::vector<uint32_t> dcol(P.mapStride, 0xFFFFFFFFu);
        for (uint32_t s2 = s0; s2 < s1; s2++)
          for (uint32_t i = 0; i < W.segs[s2].len; i++) dcol[W.segs[s2].lds + i] = dev_col(W.segs[s2].col + i);
        for (uint32_t b = 0; b < P.mapStride / 256u; b++) {
          uint32_t lo = 0xFFFFFFFFu;
          for (uint32_t i = 0; i < 256; i++) lo = std::min(lo, dcol[b * 256u + i]);
          if (lo == 0xFFFFFFFFu) lo = m->padCol;
          bb[b] = lo;
          for (uint32_t i = 0; i < 256; i++) mp[b * 256u + i] = dcol[b * 256u + i] == 0xFFFFFFFFu ? 0 : (uint16_t)(dcol[b * 256u + i] - lo);
        }
      }
    }
  }
  P.segPtr = std::move(W.segPtr), P.segs = std::move(W.segs);
  // window slots of every element: pack_slots_k on the columns in original numbering (device copy, un-remapped)
  uint32_t *origCol = nullptr, *slots = nullptr;
  HIP_CHECK(hipMalloc(&origCol, ((size_t)m->nElems + SCS_SLACK) * sizeof(uint32_t)));
  sb_d2d(origCol, m->colInd, ((size_t)m->nElems + SCS_SLACK) * sizeof(uint32_t)); // (on g.stream, like the kernels below)
  if (m->permuted)
    hipLaunchKernelGGL(remap_cols_k, dim3(stream_grid(m->nElems, 256)), dim3(256), 0, g.stream, m->nElems, m->nr, m->newToOld, origCol);
  uint32_t* dVirt = nullptr;
  if (virt) {
    dVirt = (uint32_t*)upload(haloVirt.data(), haloVirt.size() * sizeof(uint32_t));
    hipLaunchKernelGGL(remap_halo_k, dim3(stream_grid(m->nElems, 256)), dim3(256), 0, g.stream, m->nElems, m->nr, dVirt, origCol);
  }
  uint32_t* dSegPtr = (uint32_t*)upload(P.segPtr.data(), P.segPtr.size() * sizeof(uint32_t));
  TileSeg* dSegs    = (TileSeg*)upload(P.segs.data(), P.segs.size() * sizeof(TileSeg));
  std::vector<PackMeta> meta(m->nChunks);
  sb_d2h(meta.data(), m->pmeta, meta.size() * sizeof(PackMeta));
  const uint64_t groups = (uint64_t)meta.back().grp + ((meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
  HIP_CHECK(hipMalloc(&slots, (size_t)groups * 512 + 1024));
  hipLaunchKernelGGL(pack_slots_k, dim3((m->nChunks + 3) / 4), dim3(256), 0, g.stream, m->chunkPtr, m->chunkLens, origCol,
      m->val, m->pmeta, dSegPtr, dSegs, m->nChunks, m->permuted ? 0u : m->padCol /* padding: original column 0 */, slots, cpt);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  sb_free(origCol), sb_free(dSegPtr), sb_free(dVirt);
  P.slots = slots;
  bool kept = false;
  if (pattern_classes(m, P)) {
    pattern_codes(m, P);
    kept = build_masked(m, P);
  }
  sb_free(slots);
  P.free_temporaries();
  if (!kept) {
    P.free_tables(), sb_free(dSegs);
    return false;
  }
  m->mOwnsTables = true, m->mSegs = dSegs, m->mWindow = maxWin, m->mMapStride = P.mapStride;
  { // The maps are relative to their 256-slot blocks' bases (header words), so tiles whose windows look alike up to a shift
    // hold the SAME map -- on a stencil matrix all but the boundary tiles.  Store every distinct map once, and a table
    // header -> map in front of them: [u32 index per header, padded to 256 entries of 16 bits][distinct maps]; the table's
    // length in units of 256 map entries rides in the high bits of mMapStride (the stride itself is < 4096).  At HPCG 128^3
    // that is a few hundred KB instead of 18.9 MB of map per SpMV, and the map fetch finds its lines in L2.  Where hardly
    // anything repeats (SB_PACK_MAP_DEDUP=0 forces that) the table is left out and header h's map is map h, as before.
    const size_t ms = P.mapStride;
    std::vector<uint32_t> idxOfHdr(P.nTiles);
    std::vector<const uint16_t*> uniq;
    std::unordered_multimap<uint64_t, uint32_t> seen;
    for (uint32_t hI = 0; hI < P.nTiles; hI++) {
      const uint16_t* mp = P.slotMap.data() + (size_t)m->mTileOfHdr[hI] * ms;
      uint64_t h = 1469598103934665603ull;
      for (size_t i = 0; i < ms; i++) h = (h ^ mp[i]) * 1099511628211ull;
      uint32_t found = 0xFFFFFFFFu;
      auto range = seen.equal_range(h);
      for (auto it = range.first; it != range.second; ++it)
        if (memcmp(uniq[it->second], mp, ms * sizeof(uint16_t)) == 0) { found = it->second; break; }
      if (found == 0xFFFFFFFFu) {
        found = (uint32_t)uniq.size();
        uniq.push_back(mp);
        seen.emplace(h, found);
      }
      idxOfHdr[hI] = found;
    }
    const char* de    = getenv("SB_PACK_MAP_DEDUP");
    const bool dedup  = !(de && atoi(de) == 0) && uniq.size() * 4 <= (size_t)P.nTiles * 3;
    const size_t offU = dedup ? ((size_t)P.nTiles * 2 + 255) / 256 : 0; // table length in units of 256 u16
    if (offU >= (1u << 20)) SB_FATAL("slot-map table too long");
    const size_t nMaps = dedup ? uniq.size() : (size_t)P.nTiles;
    std::vector<uint16_t> buf(offU * 256 + nMaps * ms);
    if (dedup) {
      memcpy(buf.data(), idxOfHdr.data(), (size_t)P.nTiles * sizeof(uint32_t));
      for (size_t u = 0; u < nMaps; u++) memcpy(buf.data() + offU * 256 + u * ms, uniq[u], ms * sizeof(uint16_t));
    } else {
      for (uint32_t hI = 0; hI < P.nTiles; hI++)
        memcpy(buf.data() + (size_t)hI * ms, P.slotMap.data() + (size_t)m->mTileOfHdr[hI] * ms, ms * sizeof(uint16_t));
    }
    m->mSlotMap   = (uint16_t*)upload(buf.data(), buf.size() * sizeof(uint16_t));
    m->mMapStride = (uint32_t)ms | ((uint32_t)offU << 12);
    m->mBytes += 2.0 * ms * (double)nMaps + 512.0 * offU - 2.0 * ms * (double)P.nTiles; // (build_masked counted one map per tile)
    if (getenv("SB_PACK_REPORT"))
      fprintf(stderr, "sbhip pack: slot maps: %zu distinct of %u tiles (%zu entries each)%s\n", uniq.size(), P.nTiles, ms,
          dedup ? ": stored once, indexed per header" : ": stored per tile");
  }
  m->mTileOfHdr.clear(), m->mTileOfHdr.shrink_to_fit();
  sb_free(P.dRowBase), sb_free(P.dTileClass); // (mClassDict stays)
  return true;
}

// Levels 4-5 on windows in the device numbering.  Chunks per tile: 8 (two chunks per wave: one header fetch, window
// staging and barrier for twice the rows) where every 8-chunk window fits and no tile has more pairs than a class holds,
// else 4.  SB_PAT_CPT=4 keeps level 3's tiles.  true: built; level 6 on the same windows has been tried (m->mHdrs).
static bool build_level5(sb_matrix* m, const uint32_t* chunkPtr, const uint32_t* chunkLens, const uint32_t* colInd,
    const double* val, const uint32_t* oldToNewPerm, uint32_t forceCpt = 0)
{
  const char* env = getenv("SB_PACK");
  PatternPlan P;
  uint32_t* slots8 = nullptr; // device: window slots for 8-chunk tiles (temporary)
  {
    const char* ec = getenv("SB_PAT_CPT");
    P.cpt          = ec && atoi(ec) == 4 ? 4u : 8u;
    if (m->nChunks <= 4 || forceCpt == 4) P.cpt = 4;
  }
  if (P.cpt == 8) {
    TileWindows W = compute_tile_windows(m, 8, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
    if (!W.ok && getenv("SB_PACK_REPORT")) fprintf(stderr, "sbhip pack: tiles of 8 chunks: a window does not fit\n");
    if (!W.ok) P.cpt = 4;
    else {
      P.segPtr = std::move(W.segPtr), P.segs = std::move(W.segs);
      m->patWindow = W.maxWin;
      uint32_t* dSegPtr = (uint32_t*)upload(P.segPtr.data(), P.segPtr.size() * sizeof(uint32_t));
      m->patSegs        = (TileSeg*)upload(P.segs.data(), P.segs.size() * sizeof(TileSeg));
      std::vector<PackMeta> meta(m->nChunks);
      sb_d2h(meta.data(), m->pmeta, meta.size() * sizeof(PackMeta));
      const uint64_t groups = (uint64_t)meta.back().grp + ((meta.back().info & 0x7FFFFFFFu) + 3u) / 4u;
      HIP_CHECK(hipMalloc(&slots8, (size_t)groups * 512 + 1024));
      hipLaunchKernelGGL(pack_slots_k, dim3((m->nChunks + 3) / 4), dim3(256), 0, g.stream, m->chunkPtr, m->chunkLens,
          m->colInd, m->val, m->pmeta, dSegPtr, m->patSegs, m->nChunks, m->padCol, slots8, 8u);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(g.stream));
      sb_free(dSegPtr);
      P.slots = slots8;
    }
  }
  if (P.cpt == 4) { // level 3's tiles and windows
    P.slots = m->pslots;
    P.segPtr.resize((m->nChunks + 3) / 4 + 1);
    sb_d2h(P.segPtr.data(), m->tileSegPtr, P.segPtr.size() * sizeof(uint32_t));
    P.segs.resize(std::max<size_t>(P.segPtr.back(), 1));
    if (P.segPtr.back()) sb_d2h(P.segs.data(), m->tileSegs, (size_t)P.segPtr.back() * sizeof(TileSeg));
    m->patSegs = m->tileSegs, m->patWindow = m->ldsWindow;
  }
  m->patCPT   = P.cpt;
  P.nTiles    = (m->nChunks + P.cpt - 1) / P.cpt;
  m->patNTiles = P.nTiles;
  auto give_up = [&]() { // no levels 4-5
    sb_free(slots8);
    if (m->patSegs != m->tileSegs) sb_free(m->patSegs);
    m->patSegs = nullptr;
    P.free_temporaries(), P.free_tables();
  };
  if (!pattern_classes(m, P)) { // (more pairs than a class holds in some tile: try the smaller tiles before giving up)
    give_up();
    return P.cpt == 8 ? build_level5(m, chunkPtr, chunkLens, colInd, val, oldToNewPerm, 4) : false;
  }
  pattern_codes(m, P);
  pattern_rows(m, P, P.dom, (env ? atoi(env) : 5) >= 5);
  if (P.words > 0xFFFFFFFFull || P.excEntries > 0xFFFFFFFFull) { // positions are 32-bit
    give_up();
    return false;
  }
  m->rowBase = P.dRowBase, m->tileClass = P.dTileClass, m->classDict = P.dClassDict;
  // final form: L code words / expanded exception rows of the U chunks
  const uint32_t nBlocks4 = (m->nChunks + 3) / 4;
  uint32_t* dOff   = (uint32_t*)upload(P.chunkOff.data(), P.chunkOff.size() * sizeof(uint32_t));
  uint32_t* dFlags = (uint32_t*)upload(P.chunkFlags.data(), P.chunkFlags.size() * sizeof(uint32_t));
  const size_t streamBytes = (size_t)P.words * sizeof(uint32_t) + 1024;        // slack: clamped reads
  const size_t excBytes    = ((size_t)P.excEntries + 1040) * sizeof(PatEntry); // slack: 4 x 256 unconditional reads
  HIP_CHECK(hipMalloc(&m->jcodes, streamBytes));
  HIP_CHECK(hipMalloc(&m->excRows, excBytes));
  HIP_CHECK(hipMemsetAsync(m->jcodes, 0, streamBytes, g.stream));
  HIP_CHECK(hipMemsetAsync(m->excRows, 0, excBytes, g.stream));
  hipLaunchKernelGGL(pat_compact_k, dim3(nBlocks4), dim3(256), 0, g.stream, m->pmeta, P.dLanes, m->nChunks, dOff,
      dFlags, P.dExc, m->rowBase, m->tileClass, m->classDict, m->jcodes, m->excRows, P.cpt);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(g.stream));
  sb_free(dOff), sb_free(dFlags), sb_free(slots8);
  // level 6 on the same windows (the tile shape of choice only: build_patterns tries its own shapes otherwise)
  if (P.cpt == 8 || m->nChunks <= 4)
    if (build_masked(m, P)) m->mSegs = m->patSegs, m->mWindow = m->patWindow;
  P.free_temporaries();
  m->rowPats     = (PatEntry*)upload(P.rowPats.data(), P.rowPats.size() * sizeof(PatEntry));
  m->nRowPats    = (uint32_t)P.nRowPats;
  m->nPatClasses = (uint32_t)P.classes.size();
  m->patDict     = P.anyL ? 256u : 0u;
  m->patExcLds   = P.excLds;
  const size_t nSegs = pattern_headers(m, P, &m->tileHdrs, &m->patInterior);
  m->patBytes = (double)P.words * 4.0 + 16.0 * (double)P.excEntries + 2.0 * 64.0 * m->nChunks + 16.0 * nSegs +
                (double)(P.cpt == 8 ? 384 : 192) * P.nTiles + (P.anyL ? 4096.0 * P.classes.size() : 0.0) + 16.0 * P.rowPats.size();
  return true;
}

// Levels 4-6.  Level 6 does not depend on level 5: with windows in original column order (build_masked_mapped) a tile
// has few (value, slot delta) pairs whatever the sigma sort and the halo numbering did to the device columns -- a
// rank-local brick with a lower neighbour has > 255 pairs per tile in the device numbering (no levels 4-5 at all: its
// halo columns are not permuted, its own are) and 19 programs in the original one.
static void build_patterns(sb_matrix* m, const uint32_t* chunkPtr, const uint32_t* chunkLens, const uint32_t* colInd,
    const double* val, const uint32_t* oldToNewPerm)
{
  if (m->usePacked != 2 || m->nDict <= 0) return;
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 4) < 4) return;
  const bool level5 = build_level5(m, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
  if (!m->mHdrs && (m->permuted || !level5 || m->patCPT != 8)) {
    const char* ec = getenv("SB_PAT_CPT");
    const bool eight = !(ec && atoi(ec) == 4) && m->nChunks > 4;
    if (!(eight && build_masked_mapped(m, 8, chunkPtr, chunkLens, colInd, val)))
      build_masked_mapped(m, 4, chunkPtr, chunkLens, colInd, val);
  }
  // Default kernel: the masked row programs wherever they were built (round 2, stand-alone launches: 64^3 5.6 us against
  // 6.4 us for level 3 and 7.7 us for level 5; 96^3 10.0 / 18.5 / 15.8; 128^3 18.6 / - / 27.5).  Otherwise the level-5
  // kernel once the matrix is more than one round of resident workgroups (8 per CU, 4 chunks each); below that
  // everything is one dependent-latency chain and the level-3 kernel's is shorter (64^3: 46.7k vs 43.2k CG it/s;
  // 96^3: 22.8k vs 26.3k).  sb_matrix_use_packed(m, 3) selects it regardless.
  if (m->mHdrs) m->usePacked = 5;
  else if (level5) m->usePacked = (m->nChunks + 3) / 4 > (uint32_t)g.prop.multiProcessorCount * 8u ? 3 : 2;
}

#ifndef SB_LAB
// The product ships two SpMV kernels per format: the reference-layout stream and the masked row programs (level 6).
// Levels 1-5 are BUILT on the way to level 6 (value dictionary, window slots, pattern classes), but their kernels were
// measured slower at every size and live in lab builds (-DSB_LAB) only: the streams that only those kernels
// read are released here, and a matrix without row programs runs the reference-layout kernel.
static void product_modes_only(sb_matrix* m)
{
  if (m->usePacked != 5) m->usePacked = 0;
  sb_free(m->pidx), sb_free(m->pcodes), sb_free(m->pslots), sb_free(m->jcodes), sb_free(m->excRows);
  m->pidx = nullptr, m->pcodes = nullptr, m->pslots = nullptr, m->jcodes = nullptr, m->excRows = nullptr;
}
#endif

void sb_set_external_ids(const uint32_t* global_ids, uint32_t n)
{
  g_externalIds.assign(global_ids, global_ids + (global_ids ? n : 0));
}

sb_matrix* sb_scs_upload(uint32_t nr, uint32_t nc, uint32_t C, uint32_t sigma, uint32_t nChunks,
    uint32_t nElems, const uint32_t* chunkPtr, const uint32_t* chunkLens, const uint32_t* colInd,
    const double* val, const uint32_t* oldToNewPerm, const uint32_t* newToOldPerm)
{
  need_init();
  if (C == 0) SB_FATAL("SCS chunk height C must be >= 1");
  if ((uint64_t)nChunks * C < nr) SB_FATAL("SCS nChunks*C < nr");
  if (chunkPtr[nChunks] != nElems) SB_FATAL("SCS chunkPtr[nChunks] != nElems");
  for (uint32_t c = 0; c < nChunks; c++)
    if (chunkPtr[c + 1] - chunkPtr[c] != chunkLens[c] * C)
      SB_FATAL("SCS chunk %u: chunkPtr/chunkLens inconsistent", c);
  for (uint32_t k = 0; k < nElems; k++)
    if (colInd[k] >= nc) SB_FATAL("SCS colInd[%u]=%u out of range (nc=%u)", k, colInd[k], nc);
  sb_matrix* m = new sb_matrix();
  m->fmt = 1, m->nr = nr, m->nc = nc, m->C = C, m->sigma = sigma, m->nChunks = nChunks;
  m->nElems = nElems, m->nrPadded = nChunks * C, m->nnz = nElems;
  int permuted = 0;
  if (oldToNewPerm)
    for (uint32_t i = 0; i < nr; i++) {
      if (oldToNewPerm[i] >= nr) SB_FATAL("SCS oldToNewPerm[%u]=%u out of range", i, oldToNewPerm[i]);
      if (oldToNewPerm[i] != i) permuted = 1;
    }
  if (permuted && !newToOldPerm) SB_FATAL("SCS permuted matrix needs newToOldPerm");
  m->permuted  = permuted;
  m->chunkPtr  = (uint32_t*)upload(chunkPtr, ((size_t)nChunks + 1) * sizeof(uint32_t));
  m->chunkLens = (uint32_t*)upload(chunkLens, (size_t)nChunks * sizeof(uint32_t));
  // SCS_SLACK zeroed elements behind the data: the pipelined kernel prefetches up to
  // U-1 columns past a chunk's end
  HIP_CHECK(hipMalloc(&m->colInd, ((size_t)nElems + SCS_SLACK) * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&m->val, ((size_t)nElems + SCS_SLACK) * sizeof(double)));
  HIP_CHECK(hipMemset(m->colInd + nElems, 0, SCS_SLACK * sizeof(uint32_t)));
  HIP_CHECK(hipMemset(m->val + nElems, 0, SCS_SLACK * sizeof(double)));
  if (nElems) {
    HIP_CHECK(hipMemcpy(m->colInd, colInd, (size_t)nElems * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->val, val, (size_t)nElems * sizeof(double), hipMemcpyHostToDevice));
  }
  if (permuted) {
    m->oldToNew = (uint32_t*)upload(oldToNewPerm, (size_t)nr * sizeof(uint32_t));
    m->newToOld = (uint32_t*)upload(newToOldPerm, (size_t)nr * sizeof(uint32_t));
    hipLaunchKernelGGL(remap_cols_k, dim3(stream_grid(nElems, 256)), dim3(256), 0, g.stream, nElems,
        nr, m->oldToNew, m->colInd);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(g.stream));
  }
  build_packed(m, val, oldToNewPerm);
  build_lds_windows(m, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
  build_patterns(m, chunkPtr, chunkLens, colInd, val, oldToNewPerm);
#ifndef SB_LAB
  product_modes_only(m);
#endif
  tune_matrix_placement(m);
  return m;
}

// CRS: a device-private Sell-64-1 mirror that exists only for its pattern levels (pack.hip.h).
// Row sums are taken left to right exactly as src/matrix-CRS.c:46-65 does; the SKIPPAD kernel
// does not add the mirror's padding, so the result is the CRS loop's bit for bit.  Kept only
// when the pattern levels could be built; the native CRS kernel stays selectable
// (sb_matrix_use_packed(m, 0)) and is the default for small matrices.
static void build_crs_mirror(sb_matrix* m, const uint32_t* rowPtr, const uint32_t* colInd, const double* val)
{
  const char* env = getenv("SB_PACK");
  if ((env ? atoi(env) : 5) < 4 || m->nr == 0 || m->nnz == 0) return;
  { // the pattern levels need the value dictionary (<= 256 distinct bit patterns): decide that first, cheaply
    std::vector<unsigned long long> seen;
    unsigned long long last = ~0ull;
    for (uint32_t k = 0; k < m->nnz; k++) {
      unsigned long long b;
      memcpy(&b, val + k, 8);
      if (b == last) continue;
      last = b;
      if (std::find(seen.begin(), seen.end(), b) == seen.end()) {
        seen.push_back(b);
        if (seen.size() > 255) return; // (+0.0 for padding takes one entry)
      }
    }
  }
  const uint32_t nr = m->nr, nChunks = (nr + 63) / 64;
  std::vector<uint32_t> chunkLens(nChunks, 0), chunkPtr(nChunks + 1, 0);
  for (uint32_t i = 0; i < nr; i++) chunkLens[i / 64] = std::max(chunkLens[i / 64], rowPtr[i + 1] - rowPtr[i]);
  uint64_t total = 0;
  for (uint32_t c = 0; c < nChunks; c++) {
    chunkPtr[c] = (uint32_t)total;
    total += (uint64_t)chunkLens[c] * 64;
  }
  if (total > 0xFFFFFFFFull) return;
  chunkPtr[nChunks] = (uint32_t)total;
  std::vector<uint32_t> scol(total, 0u);
  std::vector<double> sval(total, 0.0);
  for (uint32_t i = 0; i < nr; i++) {
    const size_t at = (size_t)chunkPtr[i / 64] + (i % 64);
    for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; j++) {
      unsigned long long bits;
      memcpy(&bits, val + j, 8);
      // padding is (column 0, +0.0): a stored +0.0 at column 0 would be indistinguishable from it
      if (colInd[j] == 0 && bits == 0) return;
      scol[at + (size_t)(j - rowPtr[i]) * 64] = colInd[j];
      sval[at + (size_t)(j - rowPtr[i]) * 64] = val[j];
    }
  }
  g_tunePlacement = false; // (the mirror's Sell-64-1 arrays only feed its pattern levels and are freed below)
  sb_matrix* mm   = sb_scs_upload(nr, m->nc, 64, 1, nChunks, (uint32_t)total, chunkPtr.data(), chunkLens.data(),
      scol.data(), sval.data(), nullptr, nullptr);
  g_tunePlacement = true;
#ifdef SB_LAB
  const bool useless = mm->nPatClasses == 0;
#else
  const bool useless = mm->mHdrs == nullptr; // (the product multiplies through the mirror's row programs only)
#endif
  if (useless) { // no repeating patterns: the native kernel it is
    sb_matrix_free(mm);
    return;
  }
  // only the pattern levels are used (the other SCS kernels add the padding)
  sb_free(mm->val), sb_free(mm->colInd), sb_free(mm->pidx), sb_free(mm->pcodes), sb_free(mm->pslots);
  mm->val = nullptr, mm->colInd = nullptr, mm->pidx = nullptr, mm->pcodes = nullptr, mm->pslots = nullptr;
  m->mirror    = mm;
  m->usePacked = mm->usePacked >= 3 ? mm->usePacked : 0; // the same size rule as for SCS matrices
}

// ---- placement of the reference-layout stream ----------------------------------------------------------------------------
// Where val / colInd sit in device memory decides how fast the section-8d kernel streams them (round 4, tools/placement_lab.py,
// profiles/r04_two_speeds_*: the same kernel on the same box runs at 117 ... 141 us per launch at 128^3 depending on where the
// process's allocations happened to land; within a process the time follows the offset of the arrays, in steps).  So the arrays
// live in ONE slab with PLACE_SPAN bytes of play for each, at offsets the caller (or the tuner below) chooses; moving them is a
// device-to-device copy.  Number of stored elements of the stream: nElems + SCS_SLACK (SCS), nnz + 64 (CRS).
constexpr size_t PLACE_SPAN = (size_t)256 << 20;
static size_t place_elems(const sb_matrix* m) { return m->fmt == 1 ? (size_t)m->nElems + SCS_SLACK : (size_t)m->nnz + 64; }
void sb_matrix_place(sb_matrix* m, int colOffMB, int valOffMB)
{
  need_init();
  if (!m || !m->colInd || !m->val) return;
  if (m->placeTried) SB_FATAL("sb_matrix_place is a lab call: upload with SB_PLACE=0 (the upload's tuner has placed this matrix)");
  if (colOffMB < 0 || valOffMB < 0 || (size_t)colOffMB << 20 > PLACE_SPAN || (size_t)valOffMB << 20 > PLACE_SPAN)
    SB_FATAL("sb_matrix_place: offsets must lie in [0, %zu] MB", PLACE_SPAN >> 20);
  const size_t ne = place_elems(m), colBytes = ne * sizeof(uint32_t), valBytes = ne * sizeof(double);
  const size_t colRegion = ((colBytes + PLACE_SPAN + ((size_t)2 << 20) - 1) >> 21) << 21;
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (!m->slab) {
    m->slabBytes = colRegion + valBytes + PLACE_SPAN + ((size_t)2 << 20);
    HIP_CHECK(hipMalloc(&m->slab, m->slabBytes));
    m->colInd0 = m->colInd, m->val0 = m->val; // keep the first upload as the source of every placement
  }
  uint32_t* c = reinterpret_cast<uint32_t*>(m->slab + ((size_t)colOffMB << 20));
  double* v   = reinterpret_cast<double*>(m->slab + colRegion + ((size_t)valOffMB << 20));
  HIP_CHECK(hipMemcpy(c, m->colInd0, colBytes, hipMemcpyDeviceToDevice));
  HIP_CHECK(hipMemcpy(v, m->val0, valBytes, hipMemcpyDeviceToDevice));
  m->colInd = c, m->val = v, m->placeColMB = colOffMB, m->placeValMB = valOffMB;
}
// the same arrays in a NEW slab (another piece of device memory; the earlier slabs stay allocated, so the new one cannot land on
// the same pages): what sb_matrix_place cannot vary is WHICH memory the slab is made of
void sb_matrix_place_fresh(sb_matrix* m)
{
  need_init();
  if (!m || !m->colInd || !m->val) return;
  if (m->placeTried) SB_FATAL("sb_matrix_place_fresh is a lab call: upload with SB_PLACE=0 (the upload's tuner has placed this matrix)");
  HIP_CHECK(hipStreamSynchronize(g.stream));
  if (m->slab) m->oldSlabs.push_back(m->slab), m->slab = nullptr;
  else m->colInd0 = m->colInd, m->val0 = m->val;
  const size_t ne = place_elems(m), colBytes = ne * sizeof(uint32_t), valBytes = ne * sizeof(double);
  const size_t colRegion = ((colBytes + PLACE_SPAN + ((size_t)2 << 20) - 1) >> 21) << 21;
  m->slabBytes = colRegion + valBytes + PLACE_SPAN + ((size_t)2 << 20);
  HIP_CHECK(hipMalloc(&m->slab, m->slabBytes));
  uint32_t* c = reinterpret_cast<uint32_t*>(m->slab);
  double* v   = reinterpret_cast<double*>(m->slab + colRegion);
  HIP_CHECK(hipMemcpy(c, m->colInd0, colBytes, hipMemcpyDeviceToDevice));
  HIP_CHECK(hipMemcpy(v, m->val0, valBytes, hipMemcpyDeviceToDevice));
  m->colInd = c, m->val = v, m->placeColMB = 0, m->placeValMB = 0;
}
// the placement is final: the first upload and the slabs tried before go
void sb_matrix_place_commit(sb_matrix* m)
{
  if (!m || !m->slab) return;
  HIP_CHECK(hipStreamSynchronize(g.stream));
  sb_free(m->colInd0), sb_free(m->val0);
  m->colInd0 = nullptr, m->val0 = nullptr;
  for (char* q : m->oldSlabs) sb_free(q);
  m->oldSlabs.clear();
}
void sb_matrix_placement(const sb_matrix* m, int out[2]) { out[0] = m->placeColMB, out[1] = m->placeValMB; }
// what the placement tuner of the upload saw: probes timed (0: it did not run); us: proxy step of the loop with the first
// vectors' arena tried and the stream where hipMalloc put it, at the pair that was kept, at the slowest pair seen
int sb_matrix_placement_report(const sb_matrix* m, float us[3])
{
  for (int i = 0; i < 3; i++) us[i] = m->placeUs[i];
  return m->placeTried;
}
// (lab: tools/placement_lab2.py) device addresses of the streamed arrays
void sb_matrix_debug_ptrs(const sb_matrix* m, unsigned long long out[4])
{
  out[0] = (unsigned long long)m->colInd, out[1] = (unsigned long long)m->val;
  out[2] = (unsigned long long)(m->fmt == 1 ? (void*)m->chunkPtr : (void*)m->rowPtr), out[3] = (unsigned long long)m->chunkLens;
}

void sb_matrix_free(sb_matrix* m)
{
  if (!m) return;
  sb_free(m->vecArena);
  if (m->slab) { // (colInd / val point into the slab)
    sb_free(m->slab), sb_free(m->colInd0), sb_free(m->val0);
    for (char* q : m->oldSlabs) sb_free(q);
    m->colInd = nullptr, m->val = nullptr;
  }
  sb_free(m->rowPtr), sb_free(m->rowBlocks), sb_free(m->tileRow), sb_free(m->chunkPtr), sb_free(m->chunkLens);
  sb_free(m->oldToNew), sb_free(m->newToOld), sb_free(m->colInd), sb_free(m->val);
  sb_free(m->pmeta), sb_free(m->pidx), sb_free(m->pcodes), sb_free(m->pdict);
  if (m->patSegs != m->tileSegs) sb_free(m->patSegs);
  sb_free(m->tileSegPtr), sb_free(m->tileSegs), sb_free(m->pslots);
  sb_free(m->rowBase), sb_free(m->tileClass), sb_free(m->jcodes), sb_free(m->classDict), sb_free(m->tileHdrs), sb_free(m->rowPats), sb_free(m->excRows);
  sb_free(m->mHdrs), sb_free(m->mStream), sb_free(m->mRowBase), sb_free(m->mProgs), sb_free(m->mSlotMap);
  if (m->mOwnsTables) sb_free(m->mClassDict), sb_free(m->mSegs);
  if (m->mirror) sb_matrix_free(m->mirror);
  delete m;
}

int sb_matrix_pack_level(const sb_matrix* m) { return m->packLevel; }
void sb_matrix_use_packed(sb_matrix* m, int mode)
{ // 0 reference-layout stream, 1 packed stream + gathers through the cache, 2 packed + LDS window,
  // 3 pattern codes + LDS window; a mode the matrix does not have falls to the next lower one
  // (5: the pattern kernel on masked row programs, pack.hip.h level 6)
#ifdef SB_LAB
  if (m->fmt == 0) m->usePacked = mode >= 5 && m->mirror && m->mirror->mHdrs ? 5 : mode >= 3 && m->mirror ? 3 : 0;
  else if (mode >= 5 && m->mHdrs) m->usePacked = 5;
  else if (mode >= 3 && m->nPatClasses) m->usePacked = 3;
  else if (mode >= 2 && m->ldsWindow) m->usePacked = 2;
  else if (mode >= 1 && m->packLevel) m->usePacked = 1;
  else m->usePacked = 0;
#else // the product: masked row programs (5) where the matrix has them, or the reference-layout stream (0)
  const sb_matrix* pm = m->fmt == 0 ? m->mirror : m;
  m->usePacked        = mode >= 5 && pm && pm->mHdrs ? 5 : 0;
#endif
}
int sb_matrix_packed_mode(const sb_matrix* m) { return m->usePacked; }
int sb_matrix_crs_kernel(const sb_matrix* m) { return m->fmt == 0 && m->tileRow ? 1 : 0; }
// the matrix whose pattern levels serve m: m itself (SCS) or its private mirror (CRS)
static const sb_matrix* pat_of(const sb_matrix* m) { return m->fmt == 0 && m->mirror ? m->mirror : m; }
uint32_t sb_matrix_lds_window(const sb_matrix* m)
{ // doubles per workgroup of the SELECTED kernel's window (the pattern kernel may use tiles of 8 chunks)
  const sb_matrix* pm = pat_of(m);
  if (m->usePacked == 5) return pm->mWindow;
  return m->usePacked == 3 && pm->patWindow ? pm->patWindow : pm->ldsWindow;
}
uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }
uint32_t sb_matrix_row_patterns(const sb_matrix* m, uint32_t* uniformChunks)
{
  if (uniformChunks) *uniformChunks = pat_of(m)->nUniformChunks;
  return pat_of(m)->nRowPats;
}
uint32_t sb_matrix_row_programs(const sb_matrix* m, uint32_t* maskedChunks)
{
  if (maskedChunks) *maskedChunks = pat_of(m)->nMaskedChunks;
  return pat_of(m)->nProgs;
}
double sb_matrix_stream_bytes(const sb_matrix* m)
{ // bytes the SELECTED SpMV kernel moves per launch (matrix stream + x once + y once)
  if (m->fmt == 0 && m->usePacked == 5) return m->mirror->mBytes + 8.0 * m->mirror->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked == 5) return m->mBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  if (m->fmt == 0 && m->usePacked == 3) return m->mirror->patBytes + 8.0 * m->mirror->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked == 3) return m->patBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked == 2) return m->slotBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  if (m->fmt == 1 && m->usePacked) return m->packedBytes + 8.0 * m->nrPadded + 8.0 * m->nc;
  return sb_matrix_spmv_bytes(m);
}
uint32_t sb_matrix_nr(const sb_matrix* m) { return m->nr; }
uint32_t sb_matrix_nc(const sb_matrix* m) { return m->nc; }
int sb_matrix_is_permuted(const sb_matrix* m) { return m->permuted; }
double sb_matrix_spmv_bytes(const sb_matrix* m)
{
  if (m->fmt == 0)
    return 12.0 * m->nnz + 4.0 * ((double)m->nr + 1) + 8.0 * m->nr + 8.0 * m->nc;
  return 12.0 * m->nElems + 8.0 * m->nChunks + 8.0 * m->nrPadded + 8.0 * m->nc;
}
