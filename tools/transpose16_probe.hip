#include <hip/hip_runtime.h>
#include <cstdio>
// 16x16 bit-matrix transpose inside each 16-lane row: lane = row index, bit = column index
__device__ __forceinline__ unsigned dpp_xor1(unsigned x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned dpp_xor2(unsigned x) { return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ unsigned dpp_qrev(unsigned x) { return __builtin_amdgcn_mov_dpp(x, 0x1B, 0xF, 0xF, true); }   // quad_perm [3,2,1,0]
__device__ __forceinline__ unsigned dpp_hmir(unsigned x) { return __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true); }  // row_half_mirror
__device__ __forceinline__ unsigned dpp_rmir(unsigned x) { return __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true); }  // row_mirror
__device__ __forceinline__ unsigned transpose16(unsigned x, int lane) {
  unsigned y;
  y = dpp_hmir(dpp_rmir(x));  x = (lane & 8) ? (((y >> 8) & 0x00FFu) | (x & 0xFF00u)) : ((x & 0x00FFu) | ((y & 0x00FFu) << 8));
  y = dpp_qrev(dpp_hmir(x));  x = (lane & 4) ? (((y >> 4) & 0x0F0Fu) | (x & 0xF0F0u)) : ((x & 0x0F0Fu) | ((y & 0x0F0Fu) << 4));
  y = dpp_xor2(x);            x = (lane & 2) ? (((y >> 2) & 0x3333u) | (x & 0xCCCCu)) : ((x & 0x3333u) | ((y & 0x3333u) << 2));
  y = dpp_xor1(x);            x = (lane & 1) ? (((y >> 1) & 0x5555u) | (x & 0xAAAAu)) : ((x & 0x5555u) | ((y & 0x5555u) << 1));
  return x;
}
__global__ void k(const unsigned* in, unsigned* out) { out[threadIdx.x] = transpose16(in[threadIdx.x], threadIdx.x); }
int main() {
  unsigned h[64], o[64], *di, *dout;
  srand(3);
  for (int i = 0; i < 64; ++i) h[i] = rand() & 0xFFFF;
  hipMalloc(&di, 256); hipMalloc(&dout, 256);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c)
    bad += ((o[g*16 + r] >> c) & 1) != ((h[g*16 + c] >> r) & 1);
  printf("transpose mismatches: %d of 1024\n", bad);
  return 0;
}
