/* sbh_convert.c -- host half of convertMatrix for both formats: build the format's
 * arrays from the general matrix exactly as the reference lays them out, then hand
 * them to the HIP layer (sb_crs_upload / sb_scs_upload), which keeps them in HBM.
 *
 * Reference: src/matrix-CRS.c:12-44 and src/matrix-SCS.c:31-196.  The SCS layout is
 * pinned by the reference's fixtures tests/data/expected/test{0,8}_C_{1,2,4}_sigma_1.in.
 */
#define _GNU_SOURCE
#include <stdlib.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

static CG_UINT* row_lengths(const GMatrix* im)
{
  CG_UINT* n = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)im->nr + 1) * sizeof(CG_UINT));
  for (CG_UINT i = 0; i < im->nr; i++) n[i] = im->rowPtr[i + 1] - im->rowPtr[i];
  return n;
}

/* CRS: AoS Entry{col,val} -> SoA colInd / val, row pointer copied */
void sbh_layout_crs(CRSMatrix* m, GMatrix* im)
{
  m->startRow = im->startRow, m->stopRow = im->stopRow;
  m->totalNr = im->totalNr, m->totalNnz = im->totalNnz;
  m->nr = im->nr, m->nc = im->nc, m->nnz = im->nnz;
  const size_t stored = im->rowPtr[im->nr];
  m->rowPtr = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)im->nr + 1) * sizeof(CG_UINT));
  m->colInd = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, (stored + 1) * sizeof(CG_UINT));
  m->val    = (CG_FLOAT*)sbh_alloc_host(ARRAY_ALIGNMENT, (stored + 1) * sizeof(CG_FLOAT));
  memcpy(m->rowPtr, im->rowPtr, ((size_t)im->nr + 1) * sizeof(CG_UINT));
#pragma omp parallel for schedule(static)
  for (long k = 0; k < (long)stored; k++) {
    m->colInd[k] = im->entries[k].col;
    m->val[k]    = im->entries[k].val;
  }
  m->rowNnz = row_lengths(im);
  m->dev    = NULL;
}

void sbh_convert_crs(CRSMatrix* m, GMatrix* im)
{
  sbh_layout_crs(m, im);
  m->dev = sb_crs_upload(m->nr, m->nc, m->rowPtr, m->colInd, m->val);
}

/* Sell-C-sigma.
 *  - rows are padded up to a multiple of C (padded rows have length 0);
 *  - inside every window of sigma consecutive padded rows, rows are reordered by
 *    DESCENDING length, equal lengths keeping their order (the reference's stable
 *    sort, src/matrix-SCS.c:20-29,61-79) -- done here with a counting sort;
 *  - chunk c holds sorted rows c*C .. c*C+C-1; its width is its longest row;
 *    chunkPtr is the running sum of width*C (:88-117);
 *  - entry k of the row at sorted position q sits at chunkPtr[q/C] + k*C + q%C, i.e.
 *    column-major inside the chunk, original order inside the row (:164-192);
 *  - padding is column 0 with value 0.0 (:146-155);
 *  - columns keep their ORIGINAL numbering in this host layout.
 * Caller sets m->C and m->sigma beforehand; both are honoured. */
void sbh_layout_scs(SCSMatrix* m, GMatrix* im)
{
  const CG_UINT C = m->C ? m->C : 1, sigma = m->sigma ? m->sigma : 1;
  m->C = C, m->sigma = sigma;
  m->startRow = im->startRow, m->stopRow = im->stopRow;
  m->totalNr = im->totalNr, m->totalNnz = im->totalNnz;
  m->nr = im->nr, m->nc = im->nc, m->nnz = im->nnz;
  const CG_UINT nr = im->nr;
  m->nChunks       = (nr + C - 1) / C;
  m->nrPadded      = m->nChunks * C;
  const CG_UINT np = m->nrPadded;

  CG_UINT* len = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)np + 1) * sizeof(CG_UINT));
  CG_UINT maxLen = 0;
  for (CG_UINT i = 0; i < np; i++) {
    len[i] = i < nr ? im->rowPtr[i + 1] - im->rowPtr[i] : 0;
    if (len[i] > maxLen) maxLen = len[i];
  }
  /* sortedRow[q] = original (padded) row at sorted position q */
  CG_UINT* sortedRow = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)np + 1) * sizeof(CG_UINT));
  if (sigma == 1) {
    for (CG_UINT i = 0; i < np; i++) sortedRow[i] = i;
  } else {
    CG_UINT* bucket = (CG_UINT*)malloc(((size_t)maxLen + 2) * sizeof(CG_UINT));
    for (CG_UINT w = 0; w < np; w += sigma) {
      const CG_UINT end = w + sigma < np ? w + sigma : np;
      memset(bucket, 0, ((size_t)maxLen + 2) * sizeof(CG_UINT));
      for (CG_UINT i = w; i < end; i++) bucket[maxLen - len[i] + 1]++; /* key = maxLen-len: descending */
      for (CG_UINT k = 0; k <= maxLen; k++) bucket[k + 1] += bucket[k];
      for (CG_UINT i = w; i < end; i++) sortedRow[w + bucket[maxLen - len[i]]++] = i;
    }
    free(bucket);
  }

  m->chunkLens = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nChunks + 1) * sizeof(CG_UINT));
  m->chunkPtr  = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nChunks + 1) * sizeof(CG_UINT));
  size_t total = 0;
  for (CG_UINT c = 0; c < m->nChunks; c++) {
    CG_UINT width = 0;
    for (CG_UINT k = 0; k < C; k++)
      if (len[sortedRow[c * C + k]] > width) width = len[sortedRow[c * C + k]];
    m->chunkLens[c] = width;
    m->chunkPtr[c]  = (CG_UINT)total;
    total += (size_t)width * C;
  }
  if (total > 0xFFFFFFFFull) {
    fprintf(stderr, "sbh_convert_scs: %zu elements do not fit CG_UINT chunk pointers\n", total);
    exit(EXIT_FAILURE);
  }
  m->nElems               = (CG_UINT)total;
  m->chunkPtr[m->nChunks] = (CG_UINT)total;

  m->oldToNewPerm = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)nr + 1) * sizeof(CG_UINT));
  m->newToOldPerm = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)nr + 1) * sizeof(CG_UINT));
  for (CG_UINT q = 0; q < np; q++)
    if (sortedRow[q] < nr) m->oldToNewPerm[sortedRow[q]] = q;
  /* padded rows (length 0) sort behind every real row of their window, and only the
   * last window has any, so real rows always land on positions < nr */
  for (CG_UINT i = 0; i < nr; i++) m->newToOldPerm[m->oldToNewPerm[i]] = i;

  m->colInd = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, (total + 1) * sizeof(CG_UINT));
  m->val    = (CG_FLOAT*)sbh_alloc_host(ARRAY_ALIGNMENT, (total + 1) * sizeof(CG_FLOAT));
  memset(m->colInd, 0, (total + 1) * sizeof(CG_UINT));
  memset(m->val, 0, (total + 1) * sizeof(CG_FLOAT)); /* all-zero bits == 0.0 */
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)nr; i++) {
    const CG_UINT q   = m->oldToNewPerm[i];
    size_t at         = (size_t)m->chunkPtr[q / C] + q % C;
    const Entry* e    = im->entries + im->rowPtr[i];
    const CG_UINT cnt = len[i];
    for (CG_UINT k = 0; k < cnt; k++, at += C) {
      m->colInd[at] = e[k].col;
      m->val[at]    = e[k].val;
    }
  }
  free(sortedRow);
  m->rowNnz = len; /* first nr entries are the real rows */
  m->dev    = NULL;
}

void sbh_convert_scs(SCSMatrix* m, GMatrix* im)
{
  sbh_layout_scs(m, im);
  m->dev = sb_scs_upload(m->nr, m->nc, m->C, m->sigma, m->nChunks, m->nElems, m->chunkPtr,
      m->chunkLens, m->colInd, m->val, m->oldToNewPerm, m->newToOldPerm);
}
