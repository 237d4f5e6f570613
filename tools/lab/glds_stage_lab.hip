// glds_stage_lab.hip -- lab (round 4): would LDS-DMA staging (global_load_lds_dwordx4) make the window staging
// of the fused SpMV (spmv_prog_fusep: r and p_old windows -> LDS as r + beta p_old, then 27 LDS reads per row) faster than staging
// through registers?  The lab isolates the staging primitive in the kernel's regime: 4096 tiles of 512 rows (HPCG 128^3), one
// workgroup of 256 threads per tile, window = 3 plane segments of SEG doubles out of two 16.8 MB vectors that sit in L2 / Infinity
// Cache, then a consume phase of 27 LDS reads + 1 store per row and the p_new store for the tile's own rows.
//   mode 0  register staging: 16-byte loads of r and p, r + beta p formed in registers, ds_write       (what the product does)
//   mode 1  glds: r and p windows land raw in LDS (2 windows), a combine pass forms r + beta p in LDS, then consume
//   mode 2  glds: raw windows, no combine pass -- the consume phase reads both and forms r + beta p per use (2 x the LDS reads)
//   mode 3  no staging at all (consume only, LDS uninitialised): the floor of everything that is not staging
// build: hipcc --offload-arch=gfx950 -O3 -o bin/glds_stage_lab glds_stage_lab.hip      run: bin/glds_stage_lab [SEG=770] [reps=50]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int NX = 128, PLANE = NX * NX, ROWS = 512, MAXSEG = 1024;

template <int MODE>
__global__ __launch_bounds__(256) void tile_k(const double* __restrict__ r, const double* __restrict__ p, double* __restrict__ pnew,
    double* __restrict__ y, int nr, int seg, double beta)
{
  extern __shared__ double lds[]; // mode 0 / 3: 3 * seg doubles; mode 1 / 2: 2 * 3 * seg
  const int tid = threadIdx.x, tile = blockIdx.x;
  const int base = tile * ROWS;     // first row of the tile
  const int halo = (seg - ROWS) / 2; // rows in front of / behind the tile inside a plane segment (even)
  const int E = 3 * seg;             // window entries
  const int Epad = (E + 511) & ~511; // (a wave-instruction of the DMA writes 64 x 16 B: the tail pieces land in the pad)
  double* wr = lds;
  double* wp = lds + Epad;
  // global start of plane segment s (clamped into the vector; the lab only needs plausible addresses)
  auto seg_start = [&](int s) {
    long g = (long)base - halo + (long)(s - 1) * PLANE;
    if (g < 0) g = 0;
    if (g + seg > nr) g = nr - seg;
    return (int)(g & ~1L);
  };
  if (MODE == 0) {
    for (int e = tid * 2; e < E; e += 512) {
      const int s = e / seg, o = e - s * seg;
      const int g = seg_start(s) + o;
      const f64x2 a = *reinterpret_cast<const f64x2*>(r + g), b = *reinterpret_cast<const f64x2*>(p + g);
      f64x2 w;
      w.x = a.x + beta * b.x, w.y = a.y + beta * b.y;
      *reinterpret_cast<f64x2*>(wr + e) = w;
    }
    __syncthreads();
  } else if (MODE == 1 || MODE == 2) {
    // a wave-instruction writes 64 x 16 B contiguously at a wave-uniform LDS base: piece q = k * 256 + tid covers entries 2q, 2q + 1
    for (int k = 0; k * 512 < E; k++) {
      const int e = k * 512 + tid * 2;
      const int ec = e < E ? e : E - 2; // (the tail wave-instruction re-reads the last piece into a scratch slot behind the window)
      const int s = ec / seg, o = ec - s * seg;
      const int g = seg_start(s) + o;
      const int ebase = k * 512 + (tid & ~63) * 2; // wave-uniform
      __builtin_amdgcn_global_load_lds((gvoid*)(r + g), (lvoid*)(wr + ebase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid*)(p + g), (lvoid*)(wp + ebase), 16, 0, 0);
    }
    __syncthreads(); // (drains vmcnt: the DMA writes are complete)
    if (MODE == 1) {
      for (int e = tid * 2; e < E; e += 512) {
        const f64x2 a = *reinterpret_cast<f64x2*>(wr + e), b = *reinterpret_cast<f64x2*>(wp + e);
        f64x2 w;
        w.x = a.x + beta * b.x, w.y = a.y + beta * b.y;
        *reinterpret_cast<f64x2*>(wr + e) = w;
      }
      __syncthreads();
    }
  }
  // consume: 27 window reads per row (3 planes x 3 lines x 3 columns), two rows per thread; own rows' p_new go back to memory
  for (int row = tid; row < ROWS; row += 256) {
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
      for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          int o = halo + row + dy * NX + dx;
          o     = o < 0 ? 0 : (o >= seg ? seg - 1 : o);
          const int e = s * seg + o;
          const double w = MODE == 2 ? wr[e] + beta * wp[e] : wr[e];
          acc = acc + 0.037 * w;
        }
    const int g = base + row;
    const int ec = seg + halo + row;
    y[g]    = acc;
    pnew[g] = MODE == 2 ? wr[ec] + beta * wp[ec] : wr[ec];
  }
}

static hipEvent_t e0, e1;
template <typename F> static float timed(F f, int reps)
{
  f(), f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main(int argc, char** argv)
{
  const int seg  = argc > 1 ? atoi(argv[1]) & ~1 : 770;
  const int reps = argc > 2 ? atoi(argv[2]) : 50;
  if (seg < ROWS || seg > MAXSEG) { printf("SEG must be in [%d, %d]\n", ROWS, MAXSEG); return 1; }
  const int nr = NX * NX * NX, tiles = nr / ROWS;
  double *r, *p, *pn, *y;
  CK(hipMalloc(&r, nr * 8)); CK(hipMalloc(&p, nr * 8)); CK(hipMalloc(&pn, nr * 8)); CK(hipMalloc(&y, nr * 8));
  CK(hipMemset(r, 0x3f, nr * 8)); CK(hipMemset(p, 0x3f, nr * 8));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t epad = (size_t)((3 * seg + 511) & ~511), one = epad * 8, two = 2 * epad * 8;
  printf("tiles %d of %d rows, window 3 x %d doubles (%.1f KB per vector), %d launches per figure\n", tiles, ROWS, seg, 3 * seg * 8 / 1024.0, reps);
  double ref = 0.0, got = 0.0;
  for (int round = 0; round < 3; round++) {
    const float t0 = timed([&] { hipLaunchKernelGGL(tile_k<0>, dim3(tiles), dim3(256), one, 0, r, p, pn, y, nr, seg, 0.5); }, reps);
    CK(hipMemcpy(&ref, y + nr / 2 + 77, 8, hipMemcpyDeviceToHost));
    const float t1 = timed([&] { hipLaunchKernelGGL(tile_k<1>, dim3(tiles), dim3(256), two, 0, r, p, pn, y, nr, seg, 0.5); }, reps);
    CK(hipMemcpy(&got, y + nr / 2 + 77, 8, hipMemcpyDeviceToHost));
    const bool ok1 = got == ref;
    const float t2 = timed([&] { hipLaunchKernelGGL(tile_k<2>, dim3(tiles), dim3(256), two, 0, r, p, pn, y, nr, seg, 0.5); }, reps);
    CK(hipMemcpy(&got, y + nr / 2 + 77, 8, hipMemcpyDeviceToHost));
    const bool ok2 = got == ref;
    const float t3 = timed([&] { hipLaunchKernelGGL(tile_k<3>, dim3(tiles), dim3(256), one, 0, r, p, pn, y, nr, seg, 0.5); }, reps);
    // mode 0 with the LDS footprint of the glds modes (same occupancy): separates "staging primitive" from "fewer tiles per CU"
    const float t0b = timed([&] { hipLaunchKernelGGL(tile_k<0>, dim3(tiles), dim3(256), two, 0, r, p, pn, y, nr, seg, 0.5); }, reps);
    printf("round %d: register staging %.2f us | glds + combine pass %.2f us (%s) | glds, combine at use %.2f us (%s) | no staging %.2f us | "
           "register staging at the glds modes' LDS footprint %.2f us\n", round, t0, t1, ok1 ? "same result" : "RESULT DIFFERS", t2,
        ok2 ? "same result" : "RESULT DIFFERS", t3, t0b);
  }
  return 0;
}
