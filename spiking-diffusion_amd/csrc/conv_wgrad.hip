// Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose INPUT is a spike tensor, on the bf16 matrix cores
// (the training step of the denoiser, R/snn_model/vq_diffusion.py:166-187 through autograd; the
// reference runs cuDNN's fp32 weight-gradient kernels here):
//
//   gw[co][ky][kx][ci] = sum over (t, b, y, x) of  gy[t,b,co,y,x] * s[t,b,ci,y+ky-1,x+kx-1]          s in {0, 1}
//
// a GEMM with M = Cout, N = 9 Cin, K = T*B*49.  The spike operand is exact in bf16; the fp32 output gradient is split into
// THREE bf16 terms by truncation (x = hi + mid + lo exactly: 8 + 8 + 8 significant bits), so every product is exact and only
// the fp32 accumulation of v_mfma_f32_32x32x16_bf16 rounds -- an fp32 GEMM's accuracy at the bf16 rate (3 MFMAs per product
// tile: 178 GFLOP executed for the 256 -> 512 layer at B = 32).
//
// Mapping.  One workgroup = 128 output channels x 32 input channels x all 9 taps (144 accumulator registers per wave: wave w
// owns output channels 32 w .. 32 w + 31) over a slice of the T*B images (split K; the partial sums of the slices are added in
// a fixed order by a second launch: deterministic).  K runs image by image: the MFMA's two 8-wide k groups are two IMAGE ROWS
// (7 positions + one zero), four k steps cover the 7 rows + one zero row (23 % of the products are padding), so that
//   * the A fragment (gy, one output channel, one image row) is 8 consecutive floats of an LDS image kept [co][row][8], read
//     as two 16-byte vectors and split into its three bf16 terms in registers (each element is read by exactly one lane);
//   * the B fragment of tap (ky, kx) (spikes, one input channel, the row shifted by the tap) is ONE aligned 16-byte read of
//     an LDS image kept [kx][ci][row + 1][8] in bf16: three copies pre-shifted by kx, zero rows above and below.
// Images are double buffered: the next image's tiles travel global -> registers while the current one is multiplied.
#include "spk_common.h"
#include "den_common.h"
#include "../../include/spkdiff.h"

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int WG_CO = 128, WG_CI = 32, HW7 = 49;
constexpr int G_PITCH = 68;                       // floats per output channel in the gy image (64 + 4: 16-byte aligned rows, spread banks)
constexpr int G_FLOATS = WG_CO * G_PITCH;         // 34 816 B
constexpr int S_HALFS = 3 * WG_CI * 9 * 8;        // bf16 entries of the spike image: [kx][ci][row + 1][8] = 13 824 B

struct WgArgs {
  const float* gy; const float* s; float* part;
  int TB, Cout, Cin, ksplit;
};

__global__ __launch_bounds__(256, 1) void wgrad3x3_bf16_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  float* const sG = reinterpret_cast<float*>(lds);                                   // [2][G_FLOATS]
  unsigned short* const sS = reinterpret_cast<unsigned short*>(lds + 2 * G_FLOATS * 4);   // [2][S_HALFS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_ci = a.Cin / WG_CI, n_co = a.Cout / WG_CO;
  int bid = blockIdx.x;
  const int ks = bid % a.ksplit; bid /= a.ksplit;
  const int tn = bid % n_ci, tm = bid / n_ci;
  const int co0 = tm * WG_CO, ci0 = tn * WG_CI;
  const int per = (a.TB + a.ksplit - 1) / a.ksplit;
  const int i0 = ks * per, i1 = (i0 + per < a.TB) ? i0 + per : a.TB;

  // zero both buffers once: the pad row / column of the gy image and the border rows / columns of the spike image stay zero
  for (int i = tid; i < (2 * G_FLOATS * 4 + 2 * S_HALFS * 2) / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // this thread's share of an image: gy tile 49 x 128 floats (element e: position e / 128, channel e % 128), spike tile 49 x 32
  constexpr int NG = (HW7 * WG_CO + 255) / 256, NS = (HW7 * WG_CI + 255) / 256;      // 25, 7
  float rg[NG], rs[NS];
  auto fetch = [&](int img) {
    const float* g = a.gy + ((long long)img * HW7) * a.Cout + co0;
    const float* sp = a.s + ((long long)img * HW7) * a.Cin + ci0;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int e = tid + 256 * j;
      const int ec = e < HW7 * WG_CO ? e : HW7 * WG_CO - 1;
      rg[j] = g[(long long)(ec >> 7) * a.Cout + (ec & 127)];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int e = tid + 256 * j;
      const int ec = e < HW7 * WG_CI ? e : HW7 * WG_CI - 1;
      rs[j] = sp[(long long)(ec >> 5) * a.Cin + (ec & 31)];
    }
  };
  auto deposit = [&](int buf) {
    float* G = sG + buf * G_FLOATS;
    unsigned short* S = sS + buf * S_HALFS;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int e = tid + 256 * j;
      if (e < HW7 * WG_CO) {
        const int pos = e >> 7, co = e & 127;
        G[co * G_PITCH + (pos / 7) * 8 + (pos % 7)] = rg[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int e = tid + 256 * j;
      if (e < HW7 * WG_CI) {
        const int pos = e >> 5, ci = e & 31;
        const int y = pos / 7, x = pos % 7;
        const unsigned short v = rs[j] != 0.f ? (unsigned short)0x3F80u : (unsigned short)0u;
        // copy d (tap column kx = d): entry [row y + 1][xx] holds s(y, xx + d - 1)  ->  this spike lands at xx = x + 1 - d
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int xx = x + 1 - d;
          if (xx >= 0 && xx < 8) S[((d * WG_CI + ci) * 9 + (y + 1)) * 8 + xx] = v;
        }
      }
    }
  };

  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (i0 < i1) { fetch(i0); deposit(0); }
  __syncthreads();
  const int row = lane & 31, half = lane >> 5;
  for (int img = i0; img < i1; ++img) {
    const int buf = (img - i0) & 1;
    if (img + 1 < i1) fetch(img + 1);
    const float* G = sG + buf * G_FLOATS + (wave * 32 + row) * G_PITCH;
    const unsigned short* S = sS + buf * S_HALFS;
#pragma unroll
    for (int kstep = 0; kstep < 4; ++kstep) {
      const int y = 2 * kstep + half;                       // the image row of this lane's k group (7 = the zero row)
      const float4 g0 = *reinterpret_cast<const float4*>(G + y * 8), g1 = *reinterpret_cast<const float4*>(G + y * 8 + 4);
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      // x = hi + mid + lo exactly, each term a bf16 (top 16 bits of an fp32): truncate, subtract (exact), truncate, subtract
      unsigned hi[8], mi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned xb = __float_as_uint(gv[j]);
        hi[j] = xb & 0xFFFF0000u;
        const float r1 = gv[j] - __uint_as_float(hi[j]);
        mi[j] = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(mi[j]);
        lo[j] = __float_as_uint(r2);                        // at most 8 significant bits are left: the truncation is exact
      }
      v4i ah, am, al;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ah[j] = (int)((hi[2 * j] >> 16) | (hi[2 * j + 1] & 0xFFFF0000u));
        am[j] = (int)((mi[2 * j] >> 16) | (mi[2 * j + 1] & 0xFFFF0000u));
        al[j] = (int)((lo[2 * j] >> 16) | (lo[2 * j + 1] & 0xFFFF0000u));
      }
      const v8bf a_h = __builtin_bit_cast(v8bf, ah), a_m = __builtin_bit_cast(v8bf, am), a_l = __builtin_bit_cast(v8bf, al);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        int r = y + ky;                                     // LDS row of image row y + ky - 1
        r = r > 8 ? 8 : r;                                  // (only the zero row y = 7 can reach past the image: its A is zero)
        const v4i bv = *reinterpret_cast<const v4i*>(S + ((kx * WG_CI + row) * 9 + r) * 8);
        const v8bf b = __builtin_bit_cast(v8bf, bv);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b, acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b, acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b, acc[tap], 0, 0, 0);
      }
    }
    if (img + 1 < i1) deposit(buf ^ 1);
    __syncthreads();
  }
  // partial sums of this slice: part[ks][co][tap][ci]
  float* out = a.part + (long long)ks * a.Cout * 9 * a.Cin;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[((long long)co * 9 + tap) * a.Cin + ci0 + row] = acc[tap][r];
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, long long n, int ksplit) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[(long long)k * n + i];      // fixed order: deterministic
    gw[i] = s;
  }
}

}  // namespace

extern "C" long long spk_conv3x3_wgrad_ws_bytes(int TB, int Cout, int Cin) {
  if (TB <= 0 || Cout <= 0 || Cin <= 0 || (Cout % WG_CO) || (Cin % WG_CI)) return -1;
  const int tiles = (Cout / WG_CO) * (Cin / WG_CI);
  int ks = (spk_cu_count() + tiles - 1) / tiles;
  if (ks > TB) ks = TB;
  if (ks < 1) ks = 1;
  return (long long)ks * Cout * 9 * Cin * 4;
}

extern "C" int spk_conv3x3_wgrad_bf16(const float* gy_cl, const float* spikes_cl, float* ws, long long ws_bytes,
                                      float* gw_out, int TB, int H, int W, int Cout, int Cin, hipStream_t stream) {
  if (!gy_cl || !spikes_cl || !ws || !gw_out || TB <= 0) return SPK_ERR_ARG;
  if (H != 7 || W != 7 || (Cout % WG_CO) || (Cin % WG_CI)) return SPK_ERR_UNSUPPORTED;
  const int tiles = (Cout / WG_CO) * (Cin / WG_CI);
  int ks = (spk_cu_count() + tiles - 1) / tiles;
  if (ks > TB) ks = TB;
  if (ks < 1) ks = 1;
  const long long n = (long long)Cout * 9 * Cin;
  if (ws_bytes < (long long)ks * n * 4) return SPK_ERR_ARG;
  WgArgs a;
  a.gy = gy_cl; a.s = spikes_cl; a.part = ws; a.TB = TB; a.Cout = Cout; a.Cin = Cin; a.ksplit = ks;
  const size_t lds = 2 * (size_t)G_FLOATS * 4 + 2 * (size_t)S_HALFS * 2;
  hipLaunchKernelGGL(wgrad3x3_bf16_kernel, dim3(tiles * ks), dim3(256), lds, stream, a);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, stream,
                     ws, gw_out, n, ks);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
