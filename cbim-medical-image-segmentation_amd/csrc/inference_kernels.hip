// inference_kernels.hip — sliding-window inference accumulation and evaluation Dice.
//
//   k_softmax_accumulate  inference_sliding_window's per-window tail (/root/reference/inference/inference3d.py:80-86):
//                         pred = softmax(net(window), 1); pred_output[window] += pred; counter[window] += 1
//                         — one pass over the window's logits instead of a softmax plus two strided slice-adds
//   k_prob_finalize       pred_output /= counter (:88), optionally with torch.max(pred, dim=1) of validation.py:44
//   k_dice_counts         calculate_dice (/root/reference/metric/utils.py:62-82): per block of `block` voxels the
//                         integer counts (pred==c & target==c, pred==c, target==c); the float32 sums the reference
//                         forms from 0/1 masks are these integers, so the host reproduces its arithmetic bit for bit
// All HBM-bound streaming kernels on NCDHW float32 probabilities (the layout the model head emits).
#include "cbim_common.h"

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

namespace cbim {

static constexpr int INT_ = 256;

__global__ void __launch_bounds__(INT_) k_softmax_accumulate(const float* __restrict__ logits, float* __restrict__ acc,
                                                             float* __restrict__ counter, int K, int wd, int wh, int ww,
                                                             int D, int H, int W, int d0, int h0, int w0, int64_t total) {
  const int64_t ws = (int64_t)wd * wh * ww, S = (int64_t)D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * INT_ + threadIdx.x; i < total; i += (int64_t)gridDim.x * INT_) {
    int64_t b = i / ws, v = i % ws;
    int x = (int)(v % ww), y = (int)((v / ww) % wh), z = (int)(v / ((int64_t)ww * wh));
    const float* lp = logits + (size_t)b * K * ws + v;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, lp[(size_t)k * ws]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(lp[(size_t)k * ws] - m);
    const float inv = 1.f / s;
    const size_t o = ((size_t)(z + d0) * H + (y + h0)) * W + (x + w0);
    float* ap = acc + (size_t)b * K * S + o;
    for (int k = 0; k < K; ++k) ap[(size_t)k * S] += expf(lp[(size_t)k * ws] - m) * inv;
    if (counter) counter[(size_t)b * S + o] += 1.f;
  }
}

__global__ void __launch_bounds__(INT_) k_prob_finalize(float* __restrict__ acc, const float* __restrict__ counter,
                                                        int64_t* __restrict__ labels, int K, int64_t S, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * INT_ + threadIdx.x; i < total; i += (int64_t)gridDim.x * INT_) {
    int64_t b = i / S, v = i % S;
    const float c = counter ? counter[i] : 1.f;
    float* ap = acc + (size_t)b * K * S + v;
    float best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      float p = ap[(size_t)k * S] / c;
      ap[(size_t)k * S] = p;
      if (p > best) { best = p; arg = k; }      // first maximum, like torch.max
    }
    if (labels) labels[i] = arg;
  }
}

// counts[blk][c][3] (int32) for voxel blocks of `block` elements; one workgroup per block
template <typename TP, typename TT>
__global__ void __launch_bounds__(INT_) k_dice_counts(const TP* __restrict__ pred, const TT* __restrict__ target, int64_t N,
                                                      int64_t block, int C, int* __restrict__ counts) {
  CBIM_DYN_SMEM(raw);
  int* sh = (int*)raw;   // [C][3]
  for (int i = threadIdx.x; i < C * 3; i += INT_) sh[i] = 0;
  __syncthreads();
  const int64_t v0 = (int64_t)blockIdx.x * block;
  int64_t v1 = v0 + block;
  if (v1 > N) v1 = N;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += INT_) {
    int p = (int)pred[v], t = (int)target[v];
    if (p >= 0 && p < C) atomicAdd(&sh[p * 3 + 1], 1);        // integer atomics: exact, order-independent
    if (t >= 0 && t < C) atomicAdd(&sh[t * 3 + 2], 1);
    if (p == t && p >= 0 && p < C) atomicAdd(&sh[p * 3], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 3; i += INT_) counts[(size_t)blockIdx.x * C * 3 + i] = sh[i];
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + INT_ - 1) / INT_;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cbim

using namespace cbim;

extern "C" int cbim_softmax_accumulate(const float* logits, float* prob_sum, float* counter, int B, int K, int wd, int wh,
                                       int ww, int D, int H, int W, int d0, int h0, int w0, void* stream) {
  CBIM_CHECK(logits && prob_sum && B >= 1 && K >= 1, CBIM_EINVAL, "softmax_accumulate: bad arguments");
  CBIM_CHECK(d0 >= 0 && h0 >= 0 && w0 >= 0 && d0 + wd <= D && h0 + wh <= H && w0 + ww <= W, CBIM_EINVAL,
             "softmax_accumulate: window [%d,%d,%d]+[%d,%d,%d] outside [%d,%d,%d]", d0, h0, w0, wd, wh, ww, D, H, W);
  int64_t total = (int64_t)B * wd * wh * ww;
  CBIM_LAUNCH(k_softmax_accumulate, dim3(grid_for(total)), dim3(INT_), 0, (hipStream_t)stream, logits, prob_sum, counter, K, wd,
              wh, ww, D, H, W, d0, h0, w0, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_prob_finalize(float* prob_sum, const float* counter, int64_t* labels, int B, int K, int64_t S,
                                  void* stream) {
  CBIM_CHECK(prob_sum && B >= 1 && K >= 1 && S >= 1, CBIM_EINVAL, "prob_finalize: bad arguments");
  int64_t total = (int64_t)B * S;
  CBIM_LAUNCH(k_prob_finalize, dim3(grid_for(total)), dim3(INT_), 0, (hipStream_t)stream, prob_sum, counter, labels, K, S,
              total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_dice_counts(const void* pred, int pred_bytes, const void* target, int target_bytes, int64_t N,
                                int64_t block, int C, int32_t* counts, void* stream) {
  CBIM_CHECK(pred && target && counts && N >= 1 && block >= 1 && C >= 1 && C <= 1024, CBIM_EINVAL, "dice_counts: bad arguments");
  CBIM_CHECK((pred_bytes == 1 || pred_bytes == 8) && (target_bytes == 1 || target_bytes == 8), CBIM_EUNSUPPORTED,
             "dice_counts: labels must be int8 or int64");
  int nblk = (int)((N + block - 1) / block);
  size_t sh = (size_t)C * 3 * sizeof(int);
  hipStream_t st = (hipStream_t)stream;
  if (pred_bytes == 8 && target_bytes == 8)
    CBIM_LAUNCH((k_dice_counts<int64_t, int64_t>), dim3(nblk), dim3(INT_), sh, st, (const int64_t*)pred, (const int64_t*)target, N, block, C, counts);
  else if (pred_bytes == 8)
    CBIM_LAUNCH((k_dice_counts<int64_t, int8_t>), dim3(nblk), dim3(INT_), sh, st, (const int64_t*)pred, (const int8_t*)target, N, block, C, counts);
  else if (target_bytes == 8)
    CBIM_LAUNCH((k_dice_counts<int8_t, int64_t>), dim3(nblk), dim3(INT_), sh, st, (const int8_t*)pred, (const int64_t*)target, N, block, C, counts);
  else
    CBIM_LAUNCH((k_dice_counts<int8_t, int8_t>), dim3(nblk), dim3(INT_), sh, st, (const int8_t*)pred, (const int8_t*)target, N, block, C, counts);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(inference)
