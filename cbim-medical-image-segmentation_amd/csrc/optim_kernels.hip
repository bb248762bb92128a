// optim_kernels.hip — multi-tensor AdamW (+ optional EMA of the weights) in ONE launch.
//
// Replaces, per training iteration, torch.optim.AdamW(..., eps=1e-5).step() (/root/reference/training/utils.py:14,
// called at train.py:216) and update_ema_variables (training/utils.py:98-105, train.py:217-218): 45-278 tensors,
// several elementwise launches each in the eager path -> one streaming kernel, 16 bytes/parameter read
// (p, g, m, v [, ema]) and written (p, m, v [, ema]).  HBM-bound.
//
// hyper (device float[10]) = {step, lr, beta1, beta2, eps, weight_decay, ema_alpha, -, 1-beta1, 1-beta2} (the last two
// rounded from double on the host, as torch does): the step counter lives on
// the device (k_optim_tick) so the whole update is capturable in a hipGraph; the host only rewrites hyper[1]
// when the scheduler changes the learning rate.
//   AdamW (decoupled decay, torch semantics):  p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//     p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
//   EMA: a = min(1 - 1/t, alpha)  (global_step = t-1 in the reference's numbering);  ema = a*ema + (1-a)*p_new
#include "cbim_common.h"

namespace cbim {

static constexpr int ONT = 256;
static constexpr int OCHUNK = 4096;   // parameters per workgroup

__global__ void k_optim_tick(float* hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) hyper[0] += 1.0f;
}

__global__ void __launch_bounds__(ONT) k_adamw_ema(const cbim_optim_tensor* __restrict__ tab,
                                                   const int32_t* __restrict__ blk_tensor,
                                                   const int32_t* __restrict__ blk_chunk,
                                                   const float* __restrict__ hyper) {
  const cbim_optim_tensor t = tab[blk_tensor[blockIdx.x]];
  const int64_t base = (int64_t)blk_chunk[blockIdx.x] * OCHUNK;
  const float step = hyper[0], lr = hyper[1], b1 = hyper[2], b2 = hyper[3], eps = hyper[4], wd = hyper[5];
  const float alpha_max = hyper[6], omb1 = hyper[8], omb2 = hyper[9];
  const float bc1 = 1.f - powf(b1, step), bc2s = sqrtf(1.f - powf(b2, step));
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  float a = 1.f - 1.f / step;
  if (a > alpha_max) a = alpha_max;
  float* p = (float*)t.p;
  const float* g = (const float*)t.g;
  float* m = (float*)t.m;
  float* v = (float*)t.v;
  float* e = (float*)t.ema;
  for (int i = threadIdx.x; i < OCHUNK; i += ONT) {
    int64_t k = base + i;
    if (k >= t.numel) break;
    float gv = g[k], pv = p[k] * decay;
    float mv = b1 * m[k] + omb1 * gv;
    float vv = b2 * v[k] + omb2 * gv * gv;
    pv -= step_size * mv / (sqrtf(vv) / bc2s + eps);
    p[k] = pv; m[k] = mv; v[k] = vv;
    if (e) e[k] = a * e[k] + (1.f - a) * pv;
  }
}

// update_ema_variables alone (training/utils.py:98-102), for callers that keep torch's optimizer:
// ema = a*ema + (1-a)*p over every tensor of the table in one launch (records use .p and .ema only).
__global__ void __launch_bounds__(ONT) k_ema(const cbim_optim_tensor* __restrict__ tab, const int32_t* __restrict__ blk_tensor,
                                             const int32_t* __restrict__ blk_chunk, float a, float oma) {
  const cbim_optim_tensor t = tab[blk_tensor[blockIdx.x]];
  const int64_t base = (int64_t)blk_chunk[blockIdx.x] * OCHUNK;
  const float* p = (const float*)t.p;
  float* e = (float*)t.ema;
  for (int i = threadIdx.x; i < OCHUNK; i += ONT) {
    int64_t k = base + i;
    if (k >= t.numel) break;
    e[k] = fmaf(oma, p[k], e[k] * a);   // ema.mul_(alpha) rounds, .add_(param, alpha=1-alpha) is a fused multiply-add in ATen
  }
}

}  // namespace cbim

using namespace cbim;

extern "C" int cbim_optim_chunk(void) { return OCHUNK; }

extern "C" int cbim_adamw_ema_step(const cbim_optim_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk,
                                   int nblocks, float* hyper, void* stream) {
  CBIM_CHECK(tensors && blk_tensor && blk_chunk && hyper && nblocks >= 1, CBIM_EINVAL, "adamw_ema_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  CBIM_LAUNCH(k_optim_tick, dim3(1), dim3(64), 0, st, hyper);
  CBIM_LAUNCH(k_adamw_ema, dim3(nblocks), dim3(ONT), 0, st, tensors, blk_tensor, blk_chunk, (const float*)hyper);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "adamw_ema_step launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_ema_step(const cbim_optim_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk,
                             int nblocks, float alpha, float one_minus_alpha, void* stream) {
  CBIM_CHECK(tensors && blk_tensor && blk_chunk && nblocks >= 1, CBIM_EINVAL, "ema_step: bad arguments");
  CBIM_LAUNCH(k_ema, dim3(nblocks), dim3(ONT), 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, alpha, one_minus_alpha);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "ema_step launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(optim)
