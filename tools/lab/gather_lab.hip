// gather_lab.hip -- lab: the Sell-C-sigma stream (8-B value + 4-B column per element, wave per chunk, lane = row)
// WITH the x gather, to see what the dependent gather costs and what hides it.
//   variants: U (columns in flight per batch), PIPE (next batch's stream loads issued before this batch's gathers
//   are consumed), column pattern (all zero / stencil-like contiguous / scattered in a +-W window), x size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename T> __device__ __forceinline__ T ldnt(const T* p) { return __builtin_nontemporal_load(p); }

template <int U, bool GATHER>
__global__ __launch_bounds__(256) void kS(const double* __restrict__ val, const uint32_t* __restrict__ col, const double* __restrict__ x,
    double* __restrict__ y, uint32_t nChunks, uint32_t L)
{
  const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nChunks) return;
  const double* v = val + (size_t)chunk * L * 64 + lane;
  const uint32_t* c = col + (size_t)chunk * L * 64 + lane;
  double acc = 0.0;
  uint32_t j = 0;
  for (; j + U <= L; j += U) {
    double vv[U], xx[U]; uint32_t cc[U];
#pragma unroll
    for (int u = 0; u < U; u++) vv[u] = ldnt(v + (size_t)(j + u) * 64), cc[u] = ldnt(c + (size_t)(j + u) * 64);
#pragma unroll
    for (int u = 0; u < U; u++) xx[u] = GATHER ? x[cc[u]] : (double)cc[u];
#pragma unroll
    for (int u = 0; u < U; u++) acc += vv[u] * xx[u];
  }
  for (; j < L; j++) { double vv = ldnt(v + (size_t)j * 64); uint32_t cc = ldnt(c + (size_t)j * 64); acc += vv * (GATHER ? x[cc] : (double)cc); }
  y[chunk * 64 + lane] = acc;
}

// software pipeline: stream loads of batch b+1 in flight while the gathers of batch b are waited for
template <int U>
__global__ __launch_bounds__(256) void kP(const double* __restrict__ val, const uint32_t* __restrict__ col, const double* __restrict__ x,
    double* __restrict__ y, uint32_t nChunks, uint32_t L)
{
  const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nChunks) return;
  const double* v = val + (size_t)chunk * L * 64 + lane;
  const uint32_t* c = col + (size_t)chunk * L * 64 + lane;
  double acc = 0.0;
  double vv[U], vn[U], xx[U]; uint32_t cc[U], cn[U];
  const uint32_t nb = L / U; // L is a multiple of U here
#pragma unroll
  for (int u = 0; u < U; u++) vv[u] = ldnt(v + (size_t)u * 64), cc[u] = ldnt(c + (size_t)u * 64);
  for (uint32_t b = 0; b < nb; b++) {
#pragma unroll
    for (int u = 0; u < U; u++) xx[u] = x[cc[u]];
    const uint32_t jn = min((b + 1) * U, L - U); // clamped: unconditional
#pragma unroll
    for (int u = 0; u < U; u++) vn[u] = ldnt(v + (size_t)(jn + u) * 64), cn[u] = ldnt(c + (size_t)(jn + u) * 64);
#pragma unroll
    for (int u = 0; u < U; u++) acc += vv[u] * xx[u];
#pragma unroll
    for (int u = 0; u < U; u++) vv[u] = vn[u], cc[u] = cn[u];
  }
  y[chunk * 64 + lane] = acc;
}

template <typename F> double timeit(F f, int reps)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
  return ms * 1e3 / reps;
}

int main()
{
  const uint32_t nChunks = 32768, L = 28, nr = nChunks * 64;
  const size_t nEl = (size_t)nChunks * L * 64;
  std::vector<double> hv(nEl); std::vector<uint32_t> hc(nEl);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < nEl; i++) hv[i] = (double)(rnd() >> 11) * 0x1p-53 - 0.5;
  double *val, *x, *y; uint32_t* col;
  CK(hipMalloc(&val, nEl * 8)); CK(hipMalloc(&col, nEl * 4)); CK(hipMalloc(&x, (size_t)nr * 8)); CK(hipMalloc(&y, (size_t)nr * 8));
  CK(hipMemcpy(val, hv.data(), nEl * 8, hipMemcpyHostToDevice));
  std::vector<double> hx(nr); for (auto& e : hx) e = (double)(rnd() >> 11) * 0x1p-53;
  CK(hipMemcpy(x, hx.data(), (size_t)nr * 8, hipMemcpyHostToDevice));
  const double bytes = nEl * 12.0 + 16.0 * nr;
  const dim3 grid(nChunks / 4), block(256);
  const int reps = 20;
  const char* patName[] = { "cols all 0", "stencil-like (27 offsets, 128^2 planes)", "scattered +-2000", "scattered anywhere" };
  for (int pat = 0; pat < 4; pat++) {
    for (uint32_t ch = 0; ch < nChunks; ch++)
      for (uint32_t j = 0; j < L; j++)
        for (uint32_t k = 0; k < 64; k++) {
          const long row = (long)ch * 64 + k;
          long cidx = 0;
          if (pat == 1) { const int dz = (int)(j / 9) - 1, dy = (int)((j / 3) % 3) - 1, dx = (int)(j % 3) - 1; cidx = row + dz * 16384L + dy * 128L + dx; }
          else if (pat == 2) cidx = row + (long)(rnd() % 4001) - 2000;
          else if (pat == 3) cidx = (long)(rnd() % nr);
          if (cidx < 0) cidx = 0; if (cidx >= (long)nr) cidx = nr - 1;
          hc[((size_t)ch * L + j) * 64 + k] = (uint32_t)cidx;
        }
    CK(hipMemcpy(col, hc.data(), nEl * 4, hipMemcpyHostToDevice));
    printf("--- %s\n", patName[pat]);
#define RUN(name, ...) { double us = timeit([&] { __VA_ARGS__; }, reps); printf("%-28s %8.1f us  %6.0f GB/s  (%.3f)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0); }
    if (pat == 0) RUN("no gather U4", hipLaunchKernelGGL((kS<4, false>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("gather U2", hipLaunchKernelGGL((kS<2, true>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("gather U4", hipLaunchKernelGGL((kS<4, true>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("gather U7", hipLaunchKernelGGL((kS<7, true>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("gather U14", hipLaunchKernelGGL((kS<14, true>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("gather U28", hipLaunchKernelGGL((kS<28, true>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("pipelined U4", hipLaunchKernelGGL((kP<4>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("pipelined U7", hipLaunchKernelGGL((kP<7>), grid, block, 0, 0, val, col, x, y, nChunks, L));
    RUN("pipelined U14", hipLaunchKernelGGL((kP<14>), grid, block, 0, 0, val, col, x, y, nChunks, L));
  }
  return 0;
}
