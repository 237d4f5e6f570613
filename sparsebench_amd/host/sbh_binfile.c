/* sbh_binfile.c -- the SparseBench binary matrix file (.bmx), without MPI-IO.
 *
 * Format (reference: src/matrixBinfile.h:15-19, writer src/matrixBinfile.c:38-104,
 * reader :106-237): 24 header bytes "# SparseBench DataFile" (zero padded), then u32
 * totalNr, u32 totalNnz, u32 rowPtr[totalNr + 1], then totalNnz entries {u32 col; f32 val}.
 * Files written here are byte-identical to the reference's (the .bmx files in tests/golden/ref).
 *
 * The reference goes through MPI_File views; plain positional reads do the same job: every
 * rank opens the file, takes the row range the reference's sizeOfRank rule gives it
 * (src/matrixBinfile.c:15-18, :158-166), reads that slice of rowPtr and -- because rowPtr
 * is global -- finds its entries at rowPtr[startRow] without the reference's MPI_Allgather.
 *
 * Extension ("with fp64 values"): header byte 23 = '8' marks a file whose
 * entries are {u32 col; u32 0; f64 val} (the in-memory Entry), so a CG run on a reloaded
 * matrix is bit-identical to the run on the original.  Written when SB_BMX_FP64=1; the
 * reader accepts both.  A reference build reads only the f32 form.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sparsebench/sparsebench.h"

#define BMX_HEADERSIZE 24
static const char BMX_MAGIC[] = "# SparseBench DataFile"; /* 22 characters */

static void die(const char* what, const char* filename)
{
  fprintf(stderr, "ERROR: %s: %s\n", what, filename);
  exit(EXIT_FAILURE);
}

static void put(const void* p, size_t size, size_t n, FILE* f, const char* filename)
{
  if (n && fwrite(p, size, n, f) != n) die("short write to matrix file", filename);
}

static void get(void* p, size_t size, size_t n, FILE* f, const char* filename)
{
  if (n && fread(p, size, n, f) != n) die("short read from matrix file", filename);
}

void matrixBinWrite(GMatrix* m, Comm* c, char* filename)
{
  if (c->size > 1) { /* as the reference: src/matrixBinfile.c:42-45 */
    fprintf(stderr, "ERROR: Matrix writing only supported for single rank\n");
    return;
  }
  FILE* f = fopen(filename, "wb");
  if (!f) die("cannot create matrix file", filename);
  printf("Writing matrix to %s\n", filename);
  const char* e64 = getenv("SB_BMX_FP64");
  const int fp64  = e64 && atoi(e64) != 0;
  char header[BMX_HEADERSIZE];
  memset(header, 0, sizeof header);
  memcpy(header, BMX_MAGIC, sizeof BMX_MAGIC - 1);
  if (fp64) header[BMX_HEADERSIZE - 1] = '8';
  put(header, 1, BMX_HEADERSIZE, f, filename);
  const unsigned int totals[2] = { (unsigned int)m->totalNr, (unsigned int)m->totalNnz };
  put(totals, sizeof(unsigned int), 2, f, filename);
  put(m->rowPtr, sizeof(CG_UINT), (size_t)m->totalNr + 1, f, filename);
  if (fp64) {
    Entry* out = (Entry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nnz + 1) * sizeof(Entry));
    memset(out, 0, ((size_t)m->nnz + 1) * sizeof(Entry)); /* padding bytes of Entry are written too */
    for (size_t i = 0; i < m->nnz; i++) out[i].col = m->entries[i].col, out[i].val = m->entries[i].val;
    put(out, sizeof(Entry), m->nnz, f, filename);
    free(out);
  } else {
    FEntry* out = (FEntry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nnz + 1) * sizeof(FEntry));
    for (size_t i = 0; i < m->nnz; i++) out[i].col = (unsigned int)m->entries[i].col, out[i].val = (float)m->entries[i].val;
    put(out, sizeof(FEntry), m->nnz, f, filename);
    free(out);
  }
  if (fclose(f) != 0) die("cannot close matrix file", filename);
}

void matrixBinRead(GMatrix* m, Comm* c, char* filename)
{
  FILE* f = fopen(filename, "rb");
  if (!f) die("cannot open matrix file", filename);
  if (commIsMaster(c)) printf("Reading matrix from %s\n", filename);
  char header[BMX_HEADERSIZE];
  get(header, 1, BMX_HEADERSIZE, f, filename);
  if (memcmp(header, BMX_MAGIC, sizeof BMX_MAGIC - 1) != 0) die("not a SparseBench binary matrix file", filename);
  const int fp64 = header[BMX_HEADERSIZE - 1] == '8';
  unsigned int totals[2];
  get(totals, sizeof(unsigned int), 2, f, filename);
  const unsigned int totalNr = totals[0], totalNnz = totals[1];
  m->totalNr  = (CG_UINT)totalNr;
  m->totalNnz = (CG_UINT)totalNnz;
  printf("Rank %d: totalNr %u totalNnz %u\n", c->rank, m->totalNr, m->totalNnz);

  /* row-wise partition, src/matrixBinfile.c:155-166 */
  unsigned int numRows = 0, startRow = 0, cursor = 0;
  for (int i = 0; i <= c->rank; i++) {
    numRows  = totalNr / (unsigned int)c->size + ((totalNr % (unsigned int)c->size > (unsigned int)i) ? 1u : 0u);
    startRow = cursor;
    cursor += numRows;
  }
  printf("Rank %d: numRows %u startRow %u stopRow %u\n", c->rank, numRows, startRow, cursor - 1);
  m->nr       = (CG_UINT)numRows;
  m->nc       = (CG_UINT)numRows; /* as the reference; commPartition renumbers and sets nc */
  m->startRow = (CG_UINT)startRow;
  m->stopRow  = (CG_UINT)(cursor - 1);

  const long rowPtrAt = BMX_HEADERSIZE + 2 * (long)sizeof(unsigned int);
  m->rowPtr = (CG_UINT*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)numRows + 1) * sizeof(CG_UINT));
  if (fseek(f, rowPtrAt + (long)startRow * (long)sizeof(unsigned int), SEEK_SET) != 0) die("seek failed", filename);
  get(m->rowPtr, sizeof(CG_UINT), (size_t)numRows + 1, f, filename);
  const CG_UINT entryOffset = m->rowPtr[0];
  if (m->rowPtr[numRows] < entryOffset || m->rowPtr[numRows] > totalNnz) die("corrupt row pointers", filename);
  for (unsigned int i = 0; i <= numRows; i++) m->rowPtr[i] -= entryOffset;
  m->nnz = m->rowPtr[numRows];

  const long entriesAt = rowPtrAt + ((long)totalNr + 1) * (long)sizeof(unsigned int);
  m->entries = (Entry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nnz + 1) * sizeof(Entry));
  if (fp64) {
    if (fseek(f, entriesAt + (long)entryOffset * (long)sizeof(Entry), SEEK_SET) != 0) die("seek failed", filename);
    get(m->entries, sizeof(Entry), m->nnz, f, filename);
  } else {
    FEntry* in = (FEntry*)sbh_alloc_host(ARRAY_ALIGNMENT, ((size_t)m->nnz + 1) * sizeof(FEntry));
    if (fseek(f, entriesAt + (long)entryOffset * (long)sizeof(FEntry), SEEK_SET) != 0) die("seek failed", filename);
    get(in, sizeof(FEntry), m->nnz, f, filename);
    for (size_t i = 0; i < m->nnz; i++) m->entries[i].col = (CG_UINT)in[i].col, m->entries[i].val = (CG_FLOAT)in[i].val;
    free(in);
  }
  for (size_t i = 0; i < m->nnz; i++)
    if (m->entries[i].col >= totalNr) die("column index out of range in matrix file", filename);
  fclose(f);
}

/* src/main.c:54-84: where the driver's matrix comes from */
void sbh_init_matrix(Comm* c, Parameter* p, GMatrix* m)
{
  if (strcmp(p->filename, "generate") == 0) {
    matrixGenerate(m, p, c->rank, c->size, false);
    return;
  }
  if (strcmp(p->filename, "irregular") == 0) { /* Flan_1565 stand-in (sbh_irregular.c), enters like a file */
    sbh_matrix_generate_irregular(m, p, c->rank, c->size);
    return;
  }
  if (strcmp(p->filename, "generate7P") == 0) {
    matrixGenerate(m, p, c->rank, c->size, true);
    return;
  }
  const char* dot = strrchr(p->filename, '.');
  if (dot && strcmp(dot, ".mtx") == 0) {
    MMMatrix mm, local;
    memset(&mm, 0, sizeof mm), memset(&local, 0, sizeof local);
    if (commIsMaster(c)) printf("Read MTX matrix\n");
    MMMatrixRead(&mm, p->filename); /* every rank reads: no scatter needed */
    sbh_distribute_local(c, &mm, &local);
    matrixConvertfromMM(&local, m);
    free(mm.entries);
  } else if (dot && strcmp(dot, ".bmx") == 0) {
    if (commIsMaster(c)) printf("Read BMX matrix\n");
    matrixBinRead(m, c, p->filename);
  } else {
    printf("Unknown matrix file format!\n");
    commAbort(c, "Only generate, generate7P, .mtx and .bmx inputs are supported");
  }
}

/* src/util.c:11-32: the name up to its last dot (all of it if there is none) + newEnding; the
 * caller owns the malloc'ed result */
char* changeFileEnding(char* filename, char* newEnding)
{
  const char* dot  = strrchr(filename, '.');
  const size_t len = dot ? (size_t)(dot - filename) : strlen(filename);
  char* out        = (char*)malloc(len + strlen(newEnding) + 1);
  memcpy(out, filename, len);
  strcpy(out + len, newEnding);
  return out;
}

/* src/main.c:41-52: file.mtx -> file.bmx */
void sbh_write_bin_matrix(Comm* c, char* mtxFilename)
{
  MMMatrix mm, local;
  GMatrix m;
  memset(&mm, 0, sizeof mm), memset(&local, 0, sizeof local), memset(&m, 0, sizeof m);
  MMMatrixRead(&mm, mtxFilename);
  sbh_distribute_local(c, &mm, &local);
  matrixConvertfromMM(&local, &m);
  char* out = changeFileEnding(mtxFilename, ".bmx");
  matrixBinWrite(&m, c, out);
  free(out), free(mm.entries);
}
