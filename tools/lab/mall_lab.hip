// mall_lab.hip -- lab (round 3): does the 256 MiB Infinity Cache keep the CG vectors across the matrix stream?
// One CG iteration on the reference layout moves 706 MB of matrix + 134 MB of vector traffic; the five vectors are
// 84 MB.  If the matrix stream did not allocate in the Infinity Cache the vectors would stay on-die for ever: no
// write-back of p / x / r into the SpMV's time window, vector kernels at cache speed.  This lab measures what the
// memory system really does, by policy of the streaming read:
//   policy 0 plain loads, 1 non-temporal loads, 2 plain loads from hipDeviceMallocUncached memory, 3 nt from uncached
// For each policy:   write V (84 MB, dirty) | stream M (704 MB) | read V   -- times of all three, read V compared with
//   "hot" (read V right after write V) and "cold" (after 704 MB of plain stores to another buffer).
// Also: stream M after V was only READ (clean lines) vs after V was WRITTEN (dirty lines): the write-back share.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/mall_lab mall_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool NT> __global__ __launch_bounds__(256) void read_k(const f64x2* __restrict__ in, size_t n2, double* out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    f64x2 a = NT ? __builtin_nontemporal_load(in + i) : in[i], b = NT ? __builtin_nontemporal_load(in + i + stride) : in[i + stride];
    f64x2 c = NT ? __builtin_nontemporal_load(in + i + 2 * stride) : in[i + 2 * stride], d = NT ? __builtin_nontemporal_load(in + i + 3 * stride) : in[i + 3 * stride];
    acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  for (; i < n2; i += stride) { f64x2 a = in[i]; acc += a.x + a.y; }
  if (acc == 123.456) out[0] = acc;
}
template <bool NT> __global__ __launch_bounds__(256) void write_k(f64x2* out, size_t n2, double v)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    f64x2 a = { v, v + 1.0 };
    if (NT) __builtin_nontemporal_store(a, out + i); else out[i] = a;
  }
}
// read-modify-write like the CG vector kernels (p = r + beta p): reads two, writes one
template <bool NT> __global__ __launch_bounds__(256) void axpy_k(const f64x2* __restrict__ r, f64x2* p, size_t n2, double beta)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    f64x2 a = r[i], b = p[i], o;
    o.x = a.x + beta * b.x, o.y = a.y + beta * b.y;
    if (NT) __builtin_nontemporal_store(o, p + i); else p[i] = o;
  }
}

static hipEvent_t e0, e1;
template <typename F> static float timed(F f)
{
  CK(hipEventRecord(e0, 0));
  f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f;
}

int main()
{
  const size_t VB = 84ull << 20, MB = 704ull << 20;
  f64x2 *V, *W, *M, *Mu, *Junk;
  double* out;
  CK(hipMalloc(&V, VB)); CK(hipMalloc(&W, VB)); CK(hipMalloc(&M, MB)); CK(hipMalloc(&Junk, MB)); CK(hipMalloc(&out, 64));
  if (hipExtMallocWithFlags((void**)&Mu, MB, hipDeviceMallocUncached) != hipSuccess) { Mu = nullptr; (void)hipGetLastError(); printf("no uncached allocation\n"); }
  CK(hipMemset(M, 1, MB)); if (Mu) CK(hipMemset(Mu, 1, MB)); CK(hipMemset(V, 0, VB)); CK(hipMemset(W, 0, VB)); CK(hipMemset(Junk, 0, MB));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(256 * 8), block(256);
  const size_t v2 = VB / 16, m2 = MB / 16;
  auto rdV  = [&]() { hipLaunchKernelGGL(read_k<false>, grid, block, 0, 0, V, v2, out); };
  auto wrV  = [&]() { hipLaunchKernelGGL(write_k<false>, grid, block, 0, 0, V, v2, 1.0); };
  auto wrVnt = [&]() { hipLaunchKernelGGL(write_k<true>, grid, block, 0, 0, V, v2, 1.0); };
  auto junk = [&]() { hipLaunchKernelGGL(write_k<false>, grid, block, 0, 0, Junk, m2, 2.0); };
  auto stream = [&](int pol) {
    const f64x2* src = pol >= 2 ? Mu : M;
    if (pol & 1) hipLaunchKernelGGL(read_k<true>, grid, block, 0, 0, src, m2, out);
    else hipLaunchKernelGGL(read_k<false>, grid, block, 0, 0, src, m2, out);
  };
  for (int rep = 0; rep < 3; rep++) { wrV(); rdV(); stream(0); junk(); }
  CK(hipDeviceSynchronize());
  printf("V = %zu MB, M = %zu MB; times in us, rates in TB/s\n", VB >> 20, MB >> 20);
  { // hot / cold references
    junk(); wrV();
    float hot = timed(rdV);
    junk();
    float cold = timed(rdV);
    wrV(); float wr_hot = timed(wrV);
    junk(); float wr_cold = timed(wrV);
    printf("read V hot %.1f us (%.2f)   cold %.1f us (%.2f)   write V over hot %.1f (%.2f)  over cold %.1f (%.2f)\n", hot, VB / hot * 1e-6, cold,
        VB / cold * 1e-6, wr_hot, VB / wr_hot * 1e-6, wr_cold, VB / wr_cold * 1e-6);
  }
  const char* names[4] = { "plain", "nt", "uncached plain", "uncached nt" };
  for (int pol = 0; pol < 4; pol++) {
    if (pol >= 2 && !Mu) continue;
    float tS_dirty = 0, tR = 0, tS_clean = 0, tR2 = 0, tS_ntw = 0, tR3 = 0;
    for (int rep = 0; rep < 3; rep++) {
      junk(); wrV();                       // V dirty, on-die
      tS_dirty = timed([&]() { stream(pol); });
      tR = timed(rdV);                     // is V still on-die?
      junk(); wrV(); rdV(); rdV();         // hmm: still dirty; make it clean by letting junk evict it, then read it in
      junk(); rdV();                       // V clean, on-die
      tS_clean = timed([&]() { stream(pol); });
      tR2 = timed(rdV);
      junk(); wrVnt();                     // V written with nt stores
      tS_ntw = timed([&]() { stream(pol); });
      tR3 = timed(rdV);
    }
    printf("%-15s stream after dirty V %.1f us (%.2f)  then read V %.1f (%.2f) | after clean V %.1f (%.2f) then read V %.1f (%.2f) | after nt-written V %.1f (%.2f) then read V %.1f (%.2f)\n",
        names[pol], tS_dirty, MB / tS_dirty * 1e-6, tR, VB / tR * 1e-6, tS_clean, MB / tS_clean * 1e-6, tR2, VB / tR2 * 1e-6, tS_ntw,
        MB / tS_ntw * 1e-6, tR3, VB / tR3 * 1e-6);
  }
  // the vector kernels themselves: p = r + beta p with both operands hot / cold, plain and nt stores
  {
    const f64x2* r = W; f64x2* p = V;
    const size_t h2 = v2 / 5; // one CG vector = 16.8 MB
    auto ax = [&](bool nt) { if (nt) hipLaunchKernelGGL(axpy_k<true>, grid, block, 0, 0, r, p, h2, 0.5); else hipLaunchKernelGGL(axpy_k<false>, grid, block, 0, 0, r, p, h2, 0.5); };
    for (int nt = 0; nt < 2; nt++) {
      junk(); float cold = timed([&]() { ax(nt); });
      float hot = timed([&]() { ax(nt); });
      printf("axpy 3 x 16.8 MB, %s stores: cold %.1f us (%.2f TB/s)  hot %.1f us (%.2f)\n", nt ? "nt" : "plain", cold, 3.0 * h2 * 16 / cold * 1e-6, hot, 3.0 * h2 * 16 / hot * 1e-6);
    }
  }
  return 0;
}
