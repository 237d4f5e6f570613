// stream_lab.hip -- lab: what rate does a 12-byte-per-element matrix stream (8-B value + 4-B column)
// reach on MI355X as a function of the per-lane load width?  (No x gather: pure streaming.)
//   A  8 B + 4 B per lane          (spmv_scs64's loads: column-major chunk, lane = row)
//   B  16 B (4 cols) + 2 x 16 B (2 x 2 vals) per lane, every load a contiguous 1 KiB per wave
//   C  16 B per lane only          (one array: the guide's 6.3 TB/s copy-class ceiling, read side)
// build: hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// A: one wave per "chunk" of L columns x 64 rows; UNROLL columns in flight
template <bool NT, int U>
__global__ __launch_bounds__(256) void kA(const double* __restrict__ val, const uint32_t* __restrict__ col, uint32_t nChunks, uint32_t L, double* out)
{
  const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nChunks) return;
  const double* v = val + (size_t)chunk * L * 64 + lane;
  const uint32_t* c = col + (size_t)chunk * L * 64 + lane;
  double acc = 0.0;
  uint32_t j = 0;
  for (; j + U <= L; j += U) {
    double vv[U]; uint32_t cc[U];
#pragma unroll
    for (int u = 0; u < U; u++) vv[u] = ld<NT>(v + (size_t)(j + u) * 64), cc[u] = ld<NT>(c + (size_t)(j + u) * 64);
#pragma unroll
    for (int u = 0; u < U; u++) acc += vv[u] * (double)cc[u];
  }
  for (; j < L; j++) acc += ld<NT>(v + (size_t)j * 64) * (double)ld<NT>(c + (size_t)j * 64);
  if (acc == 1.2345) out[0] = acc;
}

// B: groups of 4 columns: [col block 1 KiB][val block A 1 KiB][val block B 1 KiB], lane = row
template <bool NT, int U>
__global__ __launch_bounds__(256) void kB(const char* __restrict__ mat, uint32_t nChunks, uint32_t G /*groups per chunk*/, double* out)
{
  const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nChunks) return;
  const char* base = mat + (size_t)chunk * G * 3072 + lane * 16;
  double acc = 0.0;
  uint32_t g = 0;
  for (; g + U <= G; g += U) {
    u32x4 c[U]; f64x2 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const char* p = base + (size_t)(g + u) * 3072;
      c[u] = ld<NT>((const u32x4*)p), a[u] = ld<NT>((const f64x2*)(p + 1024)), b[u] = ld<NT>((const f64x2*)(p + 2048));
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += a[u].x * (double)c[u].x + a[u].y * (double)c[u].y + b[u].x * (double)c[u].z + b[u].y * (double)c[u].w;
  }
  for (; g < G; g++) {
    const char* p = base + (size_t)g * 3072;
    u32x4 c = ld<NT>((const u32x4*)p); f64x2 a = ld<NT>((const f64x2*)(p + 1024)), b = ld<NT>((const f64x2*)(p + 2048));
    acc += a.x * (double)c.x + a.y * (double)c.y + b.x * (double)c.z + b.y * (double)c.w;
  }
  if (acc == 1.2345) out[0] = acc;
}

// C: plain 16 B/lane grid-stride read
template <bool NT>
__global__ __launch_bounds__(256) void kC(const f64x2* __restrict__ in, size_t n2, double* out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) { f64x2 v = ld<NT>(in + i); acc += v.x + v.y; }
  if (acc == 1.2345) out[0] = acc;
}
// C2: 16 B/lane, one wave per contiguous 27 KiB piece (same work shape as A/B)
template <bool NT, int U>
__global__ __launch_bounds__(256) void kC2(const char* __restrict__ mat, uint32_t nChunks, uint32_t K /*KiB per chunk*/, double* out)
{
  const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nChunks) return;
  const char* base = mat + (size_t)chunk * K * 1024 + lane * 16;
  double acc = 0.0;
  uint32_t g = 0;
  for (; g + U <= K; g += U) {
    f64x2 a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = ld<NT>((const f64x2*)(base + (size_t)(g + u) * 1024));
#pragma unroll
    for (int u = 0; u < U; u++) acc += a[u].x + a[u].y;
  }
  for (; g < K; g++) { f64x2 a = ld<NT>((const f64x2*)(base + (size_t)g * 1024)); acc += a.x + a.y; }
  if (acc == 1.2345) out[0] = acc;
}

template <typename F> double timeit(F f, int reps)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
  return ms * 1e3 / reps;
}

int main(int argc, char** argv)
{
  const uint32_t nChunks = 32768, L = 28; // 128^3-like: 58.7 M elements, 704.6 MB
  const size_t nEl = (size_t)nChunks * L * 64, bytes = nEl * 12;
  char* mat; double* out; CK(hipMalloc(&mat, bytes + 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(mat, 0, bytes + 4096));
  // a second buffer of 300 MB that is streamed between runs would flush the Infinity Cache; with 705 MB per pass it flushes itself
  const double* val = (const double*)mat; const uint32_t* col = (const uint32_t*)(mat + nEl * 8);
  const dim3 grid(nChunks / 4), block(256);
  const int reps = 20;
  printf("elements %zu, bytes %.1f MB\n", nEl, bytes / 1e6);
#define RUN(name, ...) { double us = timeit([&] { __VA_ARGS__; }, reps); printf("%-34s %8.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0); }
  RUN("A  8B+4B  plain U4", hipLaunchKernelGGL((kA<false, 4>), grid, block, 0, 0, val, col, nChunks, L, out));
  RUN("A  8B+4B  nt    U4", hipLaunchKernelGGL((kA<true, 4>), grid, block, 0, 0, val, col, nChunks, L, out));
  RUN("A  8B+4B  nt    U7", hipLaunchKernelGGL((kA<true, 7>), grid, block, 0, 0, val, col, nChunks, L, out));
  RUN("A  8B+4B  plain U7", hipLaunchKernelGGL((kA<false, 7>), grid, block, 0, 0, val, col, nChunks, L, out));
  RUN("B  16B x3 plain U1", hipLaunchKernelGGL((kB<false, 1>), grid, block, 0, 0, mat, nChunks, L / 4, out));
  RUN("B  16B x3 nt    U1", hipLaunchKernelGGL((kB<true, 1>), grid, block, 0, 0, mat, nChunks, L / 4, out));
  RUN("B  16B x3 plain U2", hipLaunchKernelGGL((kB<false, 2>), grid, block, 0, 0, mat, nChunks, L / 4, out));
  RUN("B  16B x3 nt    U2", hipLaunchKernelGGL((kB<true, 2>), grid, block, 0, 0, mat, nChunks, L / 4, out));
  RUN("B  16B x3 nt    U7", hipLaunchKernelGGL((kB<true, 7>), grid, block, 0, 0, mat, nChunks, L / 4, out));
  RUN("C2 16B wave-piece plain U4", hipLaunchKernelGGL((kC2<false, 4>), grid, block, 0, 0, mat, nChunks, L * 64 * 12 / 1024, out));
  RUN("C2 16B wave-piece nt    U4", hipLaunchKernelGGL((kC2<true, 4>), grid, block, 0, 0, mat, nChunks, L * 64 * 12 / 1024, out));
  RUN("C2 16B wave-piece nt    U7", hipLaunchKernelGGL((kC2<true, 7>), grid, block, 0, 0, mat, nChunks, L * 64 * 12 / 1024, out));
  RUN("C  16B grid-stride plain 2048 blk", hipLaunchKernelGGL((kC<false>), dim3(2048), block, 0, 0, (const f64x2*)mat, bytes / 16, out));
  RUN("C  16B grid-stride nt    2048 blk", hipLaunchKernelGGL((kC<true>), dim3(2048), block, 0, 0, (const f64x2*)mat, bytes / 16, out));
  RUN("C  16B grid-stride nt    8192 blk", hipLaunchKernelGGL((kC<true>), dim3(8192), block, 0, 0, (const f64x2*)mat, bytes / 16, out));
  return 0;
}
