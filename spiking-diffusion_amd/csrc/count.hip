// Spike counts from the tensors the kernels emit (the firing-rate input of the syops report,
// R/syops/ops.py:14-24 `spike_rate`): one pass over a spike tensor in any of the library's storage formats, all time steps
// and time step 0 alone (the reference's LIF hook reads the rate of output[0], R/syops/ops.py:69-75).
//   kind 0: u8 {0,1} bytes   ([...][T][C] "PTC" / "CPTC");   kind 1: fp4 nibbles 0x0 / 0x2 ("C4", "S32");
//   kind 2: fp32 words ([T][N], the reference interface) -- also counts the words equal to 1.0f, so that the caller can
//           tell a binary tensor (nonzero == ones) from an analogue one, as spike_rate's unique() test does.
// The tensor is read as u32 words; word w belongs to time step (w / inner_words) % T.
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {
__global__ __launch_bounds__(256) void count_spikes_kernel(const unsigned* __restrict__ d, long long n_words, long long inner_words,
                                                           int T, int kind, unsigned long long* __restrict__ out) {
  unsigned long long tot = 0, t0 = 0, ones = 0;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (long long)gridDim.x * blockDim.x) {
    const unsigned x = d[w];
    unsigned c;
    if (kind == 0) c = __popc(x & 0x01010101u);
    else if (kind == 1) c = __popc(x & 0x22222222u);
    else { c = (x & 0x7fffffffu) != 0u; ones += (x == 0x3f800000u); }
    tot += c;
    if ((w / inner_words) % T == 0) t0 += c;
  }
  __shared__ unsigned long long s[3][256];
  s[0][threadIdx.x] = tot; s[1][threadIdx.x] = t0; s[2][threadIdx.x] = ones;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k)
      for (int j = 0; j < 3; ++j) s[j][threadIdx.x] += s[j][threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x < 3 && s[threadIdx.x][0]) atomicAdd(out + threadIdx.x, s[threadIdx.x][0]);
}

// Content checksum of a set of tensors (parameters + buffers of a module) in ONE launch: the host layer compares it with the
// value its derived weight forms (digit planes, folded BN terms, captured graphs) were built from -- writes through `.data`
// and graph-replayed optimizer steps change the bytes without changing any (pointer, version) pair.
// Tensor j contributes sum_w (word_w + 1) * mix(j, w) (64-bit wrap-around: order independent, so atomics may land in any order).
__global__ __launch_bounds__(256) void checksum_multi_kernel(const unsigned long long* __restrict__ table, int n,
                                                             unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (int j = blockIdx.y; j < n; j += gridDim.y) {
    const unsigned* d = reinterpret_cast<const unsigned*>(table[2 * j]);
    const long long n_words = (long long)table[2 * j + 1];
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (long long)gridDim.x * blockDim.x) {
      const unsigned long long m = ((unsigned long long)w + 0x9E3779B97F4A7C15ull * (unsigned long long)(j + 1)) * 0xD1342543DE82EF95ull;
      acc += ((unsigned long long)d[w] + 1ull) * (m | 1ull);
    }
  }
  __shared__ unsigned long long s[256];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0 && s[0]) atomicAdd(out, s[0]);
}
}  // namespace

extern "C" int spk_checksum_multi(const unsigned long long* table_dev, int n, unsigned long long* out1, hipStream_t stream) {
  if (!table_dev || !out1 || n <= 0) return SPK_ERR_ARG;
  hipError_t e = hipMemsetAsync(out1, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(checksum_multi_kernel, dim3(128, n < 32 ? n : 32), dim3(256), 0, stream, table_dev, n, out1);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_count_spikes(const void* data, long long n_words, long long inner_words, int T, int kind,
                                unsigned long long* out3, hipStream_t stream) {
  if (!data || !out3 || n_words <= 0 || inner_words <= 0 || T <= 0 || kind < 0 || kind > 2) return SPK_ERR_ARG;
  hipError_t e = hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return (int)e;
  int blocks = spk_blocks(n_words, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(count_spikes_kernel, dim3(blocks), dim3(256), 0, stream, (const unsigned*)data, n_words, inner_words, T, kind, out3);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
