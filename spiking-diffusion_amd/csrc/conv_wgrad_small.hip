// Weight (and bias) gradient of a 3x3 / stride 1 / pad 1 convolution with FEW input channels and a dense fp32 input -- the
// denoiser's first layer in training: cat(x_t, t) has two channels (R/snn_model/vq_diffusion.py:161-165,195-201; the gradient
// autograd takes through layer.Conv2d in the training loop of R/main.py:226-252, cuDNN's weight-gradient kernel in the reference).
//   gw[co][ky][kx][ci] = sum_{n, y, x} gy[n][y][x][co] * in[n][ci][y + ky - 1][x + kx - 1],   gb[co] = sum gy[n][y][x][co]
// A GEMM with 9 * Cin = 18 columns has nothing for the matrix cores: one wave owns a contiguous run of output rows (n, y, x),
// lane = output channel (a coalesced 256-byte read of gy per row), the 9 * Cin input values of a row are wave-uniform, the
// 9 * Cin + 1 running sums stay in registers (fp32), the four waves of a workgroup meet in LDS, and a second launch adds the
// workgroups' partial sums in a fixed order in fp64: deterministic.
#include "spk_common.h"
#include "den_common.h"
#include "../../include/spkdiff.h"

namespace {

constexpr int WS_MAXHW = 64;            // 7x7 and 8x8 maps

// One wave = one image at a time (images wave, wave + W, ...): the image's Cin * HW inputs go to LDS once, its gy rows are
// requested eight at a time (independent loads), every row adds g * in[ci][tap] to 9 * Cin running sums per lane (lane = output
// channel); the four waves of a workgroup then add their sums in LDS in a fixed order: one partial set per workgroup.
template <int CIN>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const float* __restrict__ gy, const float* __restrict__ in,
                                                          float* __restrict__ part, int N, int H, int W, int Cout) {
  constexpr int K = CIN * 9;
  constexpr int PPMAX = 100;                             // (8 + 2) x (8 + 2): the image with a zero border, so that every tap is an
  __shared__ float s_in[4][CIN * PPMAX];                 //  unconditional read at a constant offset (a read under a condition compiles
                                                         //  to a branch with a wait of its own: 18 per row made the launch 45 us)
  __shared__ float s_red[3][K + 1][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cob = blockIdx.y, co = cob * 64 + lane;
  const int HW = H * W;
  float acc[K + 1];
#pragma unroll
  for (int k = 0; k <= K; ++k) acc[k] = 0.f;
  const int PW = W + 2, PP = (H + 2) * PW;
  for (int i = lane; i < CIN * PPMAX; i += 64) s_in[wv][i] = 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int n = blockIdx.x * 4 + wv; n < N; n += gridDim.x * 4) {
    for (int i = lane; i < CIN * HW; i += 64) {
      const int ci = i / HW, p = i - ci * HW, y = p / W, x = p - y * W;
      s_in[wv][ci * PP + (y + 1) * PW + x + 1] = in[(long long)n * CIN * HW + i];
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int RB = 32;                               // gy rows requested together (one memory latency per half image)
    for (int p0 = 0; p0 < HW; p0 += RB) {
      float g[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int pc = p0 + u < HW ? p0 + u : HW - 1;      // (unconditional load from a clamped row + select)
        const float ld = gy[((long long)n * HW + pc) * Cout + (co < Cout ? co : Cout - 1)];
        g[u] = (p0 + u < HW && co < Cout) ? ld : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int p = p0 + u;
        if (p < HW) {
          const int y = p / W, x = p - y * W;
          acc[K] += g[u];
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
                const float v = s_in[wv][ci * PP + (y + ky) * PW + x + kx];        // (wave-uniform address: LDS broadcast)
                acc[ci * 9 + ky * 3 + kx] = fmaf(g[u], v, acc[ci * 9 + ky * 3 + kx]);
              }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (wv > 0) {
#pragma unroll
    for (int k = 0; k <= K; ++k) s_red[wv - 1][k][lane] = acc[k];
  }
  __syncthreads();
  if (wv == 0) {
    float* o = part + (((long long)blockIdx.x * gridDim.y + cob) * (K + 1)) * 64 + lane;
#pragma unroll
    for (int k = 0; k <= K; ++k) o[k * 64] = ((acc[k] + s_red[0][k][lane]) + s_red[1][k][lane]) + s_red[2][k][lane];
  }
}

// one workgroup per (k, block of 64 channels): lane = channel, the four waves each add a quarter of the partial sets (eight
// independent chains, fixed order) in fp64 and meet in LDS in a fixed order
__global__ __launch_bounds__(256) void wgrad_small_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
                                                                 float* __restrict__ gb, int nparts, int ncob, int K, int CIN,
                                                                 int Cout, int w_channels_last) {
  __shared__ double sh[4][64];
  const int k = blockIdx.x, cob = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int co = cob * 64 + lane;
  const int per = (nparts + 3) / 4, w0 = wv * per, w1 = w0 + per < nparts ? w0 + per : nparts;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = w0; w < w1; w += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (w + u < w1) a[u] += (double)part[(((long long)(w + u) * ncob + cob) * (K + 1) + k) * 64 + lane];
  }
  sh[wv][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (wv != 0 || co >= Cout) return;
  const double s = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
  if (k == K) { if (gb) gb[co] = (float)s; return; }
  const int ci = k / 9, tap = k - 9 * ci;
  if (w_channels_last) gw[((long long)co * 9 + tap) * CIN + ci] = (float)s;      // storage [Cout][3][3][Cin]
  else gw[((long long)co * CIN + ci) * 9 + tap] = (float)s;                      // storage [Cout][Cin][3][3]
}

inline int small_parts(int N) {          // workgroups of four waves (four images at a time)
  const int want = (N + 3) / 4, cap = spk_cu_count();
  return want < cap ? want : cap;
}

}  // namespace

extern "C" long long spk_conv3x3_wgrad_small_ws_bytes(int N, int H, int W, int Cout, int Cin) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0 || Cin > 4 || H * W > WS_MAXHW) return -1;
  const long long ncob = (Cout + 63) / 64;
  return (long long)small_parts(N) * ncob * (Cin * 9 + 1) * 64 * 4;
}

extern "C" int spk_conv3x3_wgrad_small(const float* gy_cl, const float* in_nchw, float* ws, long long ws_bytes, float* gw_out,
                                       float* gb_out_or_null, int N, int H, int W, int Cout, int Cin, int weight_channels_last,
                                       hipStream_t stream) {
  if (!gy_cl || !in_nchw || !ws || !gw_out || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0) return SPK_ERR_ARG;
  if (Cin > 4 || H * W > WS_MAXHW || (H + 2) * (W + 2) > 100) return SPK_ERR_UNSUPPORTED;
  if (ws_bytes < spk_conv3x3_wgrad_small_ws_bytes(N, H, W, Cout, Cin)) return SPK_ERR_ARG;
  const int nparts = small_parts(N), ncob = (Cout + 63) / 64;
  const dim3 grid((unsigned)nparts, (unsigned)ncob);
#define SPK_WS_LAUNCH(CIN) hipLaunchKernelGGL(wgrad_small_kernel<CIN>, grid, dim3(256), 0, stream, gy_cl, in_nchw, ws, N, H, W, Cout)
  switch (Cin) {
    case 1: SPK_WS_LAUNCH(1); break;
    case 2: SPK_WS_LAUNCH(2); break;
    case 3: SPK_WS_LAUNCH(3); break;
    default: SPK_WS_LAUNCH(4); break;
  }
#undef SPK_WS_LAUNCH
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_small_reduce_kernel, dim3((unsigned)(Cin * 9 + 1), (unsigned)ncob), dim3(256), 0, stream, ws, gw_out,
                     gb_out_or_null, nparts, ncob, Cin * 9, Cin, Cout, weight_channels_last);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
