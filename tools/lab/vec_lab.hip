// vec_lab.hip -- lab (round 3): how fast can the CG vector kernels run when their operands are Infinity-Cache resident
// (the default path's whole working set, 60 MB mirror + 84 MB of vectors, fits the 256 MiB cache)?
// Shape of cg_update_p: read r, p, x (16.8 MB each), write p, x.  Sweep: threads per workgroup, workgroups per CU,
// 16-B loads in flight per thread and array (U), plain vs grid-stride vs blocked mapping.  All operands hot.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o bin/vec_lab vec_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

// p = r + beta p ; x = x + alpha p_old.  U pairs in flight per array; BLOCKED: a workgroup owns a contiguous range
template <int U, bool BLOCKED>
__global__ void pupd_k(const f64x2* __restrict__ r, f64x2* p, f64x2* x, size_t n2, double alpha, double beta)
{
  const size_t nT = (size_t)gridDim.x * blockDim.x;
  size_t i, step, end;
  if (BLOCKED) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x, step = blockDim.x, end = min(n2, (size_t)(blockIdx.x + 1) * per);
  } else {
    i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = nT, end = n2;
  }
  for (; i < end; i += (size_t)U * step) {
    f64x2 a[U], b[U], c[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = min(i + (size_t)u * step, end - 1);
      a[u] = r[j], b[u] = p[j], c[u] = x[j];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = i + (size_t)u * step;
      if (j < end) {
        f64x2 xo, po;
        xo.x = c[u].x + alpha * b[u].x, xo.y = c[u].y + alpha * b[u].y;
        po.x = a[u].x + beta * b[u].x, po.y = a[u].y + beta * b[u].y;
        x[j] = xo, p[j] = po;
      }
    }
  }
}
// r -= alpha Ap (read 2, write 1)
template <int U, bool BLOCKED>
__global__ void rupd_k(const f64x2* __restrict__ ap, f64x2* r, size_t n2, double alpha)
{
  const size_t nT = (size_t)gridDim.x * blockDim.x;
  size_t i, step, end;
  if (BLOCKED) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x, step = blockDim.x, end = min(n2, (size_t)(blockIdx.x + 1) * per);
  } else {
    i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = nT, end = n2;
  }
  for (; i < end; i += (size_t)U * step) {
    f64x2 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = min(i + (size_t)u * step, end - 1);
      a[u] = ap[j], b[u] = r[j];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = i + (size_t)u * step;
      if (j < end) {
        f64x2 o;
        o.x = b[u].x - alpha * a[u].x, o.y = b[u].y - alpha * a[u].y;
        r[j] = o;
      }
    }
  }
}

static hipEvent_t e0, e1;
template <typename F> static float timed(F f, int reps)
{
  f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main()
{
  const size_t n = 2097152, n2 = n / 2, VB = n * 8;
  f64x2 *r, *p, *x, *ap, *mirror;
  CK(hipMalloc(&r, VB)); CK(hipMalloc(&p, VB)); CK(hipMalloc(&x, VB)); CK(hipMalloc(&ap, VB)); CK(hipMalloc(&mirror, 60u << 20));
  CK(hipMemset(r, 0, VB)); CK(hipMemset(p, 0, VB)); CK(hipMemset(x, 0, VB)); CK(hipMemset(ap, 0, VB)); CK(hipMemset(mirror, 0, 60u << 20));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("p update (read 3 x 16.8 MB, write 2 x 16.8 MB = 83.9 MB) and r update (50.3 MB), operands cache-resident; us per launch, back to back (incl. ~1.5 us boundary)\n");
#define RUNP(U, BL, TH, PERCU)                                                                                                    \
  {                                                                                                                                 \
    const int grid = 256 * PERCU;                                                                                                   \
    float t = timed([&]() { hipLaunchKernelGGL((pupd_k<U, BL>), dim3(grid), dim3(TH), 0, 0, r, p, x, n2, 0.5, 0.25); }, 50);       \
    float t2 = timed([&]() { hipLaunchKernelGGL((rupd_k<U, BL>), dim3(grid), dim3(TH), 0, 0, ap, r, n2, 0.5); }, 50);              \
    printf("U %d %-8s threads %4d wg/CU %2d : p update %6.2f us (%5.2f TB/s)   r update %6.2f us (%5.2f TB/s)\n", U, BL ? "blocked" : "strided", TH, \
        PERCU, t, 5.0 * VB / t * 1e-6, t2, 3.0 * VB / t2 * 1e-6);                                                                   \
  }
  RUNP(1, false, 1024, 2) RUNP(2, false, 1024, 2) RUNP(4, false, 1024, 2) RUNP(2, false, 1024, 1) RUNP(4, false, 1024, 1)
  RUNP(1, false, 256, 8) RUNP(2, false, 256, 8) RUNP(4, false, 256, 8) RUNP(2, false, 256, 4) RUNP(4, false, 256, 4) RUNP(8, false, 256, 4)
  RUNP(2, false, 512, 4) RUNP(4, false, 512, 4) RUNP(2, false, 512, 2) RUNP(4, false, 512, 2)
  RUNP(2, true, 1024, 2) RUNP(4, true, 1024, 2) RUNP(2, true, 256, 8) RUNP(4, true, 256, 8) RUNP(4, true, 512, 4)
  RUNP(1, false, 256, 16) RUNP(2, false, 256, 16) RUNP(1, false, 1024, 4)
  return 0;
}
