// butterfly_lab.hip -- is the cross-lane xor butterfly of the canonical dot (level 0: v += lane[i ^ off], off = 1..32)
// reproduced bit for bit by DPP moves (off 1, 2, 4, 8) and gfx950's v_permlane16/32_swap (off 16, 32) instead of
// ds_bpermute (what __shfl_xor compiles to: 12 LDS-pipe round trips per 64-bit butterfly)?  Prints per step whether
// every lane got its partner's value, for both directions of the row shifts, and times the two forms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
template <int CTRL, int ROWMASK, int BANKMASK> __device__ __forceinline__ double dppmov(double old, double v)
{
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v), o = __builtin_bit_cast(unsigned long long, old);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)b, CTRL, ROWMASK, BANKMASK, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(b >> 32), CTRL, ROWMASK, BANKMASK, false);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <int W> __device__ __forceinline__ double swap_add(double v)
{ // A' + B' of v_permlane{16,32}_swap(v, v): own + partner in every lane (fp add commutes bit for bit)
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  auto lo = W == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false)
                    : __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
  auto hi = W == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false)
                    : __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  const double A = __builtin_bit_cast(double, (unsigned long long)lo[0] | ((unsigned long long)hi[0] << 32));
  const double B = __builtin_bit_cast(double, (unsigned long long)lo[1] | ((unsigned long long)hi[1] << 32));
  return A + B;
}
__global__ void steps(const double* in, double* out)
{ // out[s][lane]: own + partner for step s by the candidate instruction; out[8 + s]: by __shfl_xor
  const double v = in[threadIdx.x];
  out[0 * 64 + threadIdx.x] = v + dppmov<0xB1, 0xF, 0xF>(v, v);
  out[1 * 64 + threadIdx.x] = v + dppmov<0x4E, 0xF, 0xF>(v, v);
  { double p = dppmov<0x104, 0xF, 0x5>(v, v); p = dppmov<0x114, 0xF, 0xA>(p, v); out[2 * 64 + threadIdx.x] = v + p; }
  { double p = dppmov<0x114, 0xF, 0x5>(v, v); p = dppmov<0x104, 0xF, 0xA>(p, v); out[6 * 64 + threadIdx.x] = v + p; } // other direction
  out[3 * 64 + threadIdx.x] = v + dppmov<0x128, 0xF, 0xF>(v, v);
  out[4 * 64 + threadIdx.x] = swap_add<16>(v);
  out[5 * 64 + threadIdx.x] = swap_add<32>(v);
  for (int s = 0; s < 6; s++) out[(8 + s) * 64 + threadIdx.x] = v + __shfl_xor(v, 1 << s, 64);
}
__device__ __forceinline__ double bf_dpp(double v)
{
  v = v + dppmov<0xB1, 0xF, 0xF>(v, v);
  v = v + dppmov<0x4E, 0xF, 0xF>(v, v);
  { double p = dppmov<0x104, 0xF, 0x5>(v, v); p = dppmov<0x114, 0xF, 0xA>(p, v); v = v + p; }
  v = v + dppmov<0x128, 0xF, 0xF>(v, v);
  v = swap_add<16>(v);
  return swap_add<32>(v);
}
__device__ __forceinline__ double bf_shfl(double v)
{
  for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
template <int FORM> __global__ void timed(const double* in, double* out, int reps)
{
  double v = in[threadIdx.x + blockIdx.x * blockDim.x], acc = 0.0;
  for (int r = 0; r < reps; r++) {
    const double t = FORM ? bf_dpp(v) : bf_shfl(v);
    acc += t, v = v * 1.0000001 + 1e-9;
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
}
int main()
{
  double h[64], o[14 * 64];
  srand(7);
  for (int i = 0; i < 64; i++) h[i] = (double)rand() / RAND_MAX - 0.5 + 1e-9 * rand();
  double *din, *dout;
  hipMalloc(&din, 1 << 20), hipMalloc(&dout, 1 << 20);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(steps, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  const char* names[7] = { "xor 1  quad_perm", "xor 2  quad_perm", "xor 4  row_shl:4 banks 0,2 + row_shr:4 banks 1,3", "xor 8  row_ror:8",
    "xor 16 v_permlane16_swap", "xor 32 v_permlane32_swap", "xor 4  (the other direction)" };
  for (int s = 0; s < 7; s++) {
    const int ref = s == 6 ? 2 : s;
    int bad = 0;
    for (int l = 0; l < 64; l++) bad += memcmp(&o[s * 64 + l], &o[(8 + ref) * 64 + l], 8) != 0;
    printf("%-52s %s (%d lanes differ)\n", names[s], bad ? "WRONG" : "ok", bad);
  }
  // whole butterfly, and timing
  double big[64 * 256];
  for (int i = 0; i < 64 * 256; i++) big[i] = (double)rand() / RAND_MAX - 0.5;
  hipMemcpy(din, big, sizeof big, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int form = 0; form < 2; form++) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (form) hipLaunchKernelGGL(timed<1>, dim3(64), dim3(256), 0, 0, din, dout + form * 65536, 2000);
      else hipLaunchKernelGGL(timed<0>, dim3(64), dim3(256), 0, 0, din, dout + form * 65536, 2000);
      hipEventRecord(b), hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    printf("%s: %.1f us for 2000 butterflies per wave (%.1f ns each)\n", form ? "DPP + permlane swap" : "__shfl_xor (ds_bpermute)", 1e3 * ms, 1e6 * ms / 2000);
  }
  static double r0[64 * 256], r1[64 * 256];
  hipMemcpy(r0, dout, sizeof r0, hipMemcpyDeviceToHost), hipMemcpy(r1, dout + 65536, sizeof r1, hipMemcpyDeviceToHost);
  printf("whole butterfly, 16384 lanes x 2000 rounds: %s\n", memcmp(r0, r1, sizeof r0) ? "DIFFERENT" : "bit-identical");
  return 0;
}
