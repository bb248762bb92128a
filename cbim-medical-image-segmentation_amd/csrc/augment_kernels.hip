// augment_kernels.hip — on-device 3-D augmentation (the input pipeline stays in HBM).
//
// Reference: /root/reference/training/augmentation.py, applied per sample to tensor_img[1,C,D,H,W]
// fp32 / tensor_lab[1,1,D,H,W] in the dataset's __getitem__ (training/dataset/dim3/dataset_amos_ct.py:
// 121-153).  The reference runs ~20 ATen launches per op (affine_grid + 2x grid_sample + casts; min /
// max / mean / std / pow ...); here each op is 1-3 streaming kernels in the caller's NCDHW layout:
//   k_affine_sample3d : F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=True) for the image
//                       and grid_sample(nearest) for the label (augmentation.py:283-289) in one pass,
//                       optionally fused with the centre crop that follows it (crop_3d, :320-343)
//   k_crop3d          : crop_3d for image + label
//   k_chan_stats      : per-channel min / max / mean / unbiased std (gamma :117-124, contrast :150-155)
//   k_intensity       : brightness_multiply/additive (:67-101), gamma (:126), re-standardisation (:128-130),
//                       contrast + clamp (:158-161), gaussian_noise add (:15-17)
//   k_blur_axis       : gaussian_blur (:46-64) — the dense normalised k^3 Gaussian with zero padding is
//                       exactly separable, three 1-D passes
// All HBM-bound; random parameters are drawn on the host in the reference's order (seed parity).
#include "cbim_common.h"

namespace cbim {

static constexpr int NT = 256;

struct AffineParams {
  const float* img; const void* lab; float* oimg; int64_t* olab;
  int C, Di, Hi, Wi, Do, Ho, Wo, od0, oh0, ow0, lab_bytes;
  float th[12];
};

// torch.linspace(-1, 1, n)[i]  (symmetric evaluation, as ATen does)
__device__ __forceinline__ float lin11(int i, int n) {
  if (n <= 1) return -1.f;
  float step = 2.f / (float)(n - 1);
  return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ void __launch_bounds__(NT) k_affine_sample3d(AffineParams p) {
  const int64_t total = (int64_t)p.Do * p.Ho * p.Wo;
  const int64_t Sin = (int64_t)p.Di * p.Hi * p.Wi;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int w = (int)(i % p.Wo);
    int64_t r = i / p.Wo;
    int h = (int)(r % p.Ho), d = (int)(r / p.Ho);
    float x = lin11(p.ow0 + w, p.Wi), y = lin11(p.oh0 + h, p.Hi), z = lin11(p.od0 + d, p.Di);
    float gx = p.th[0] * x + p.th[1] * y + p.th[2] * z + p.th[3];     // theta row 0 acts on W
    float gy = p.th[4] * x + p.th[5] * y + p.th[6] * z + p.th[7];
    float gz = p.th[8] * x + p.th[9] * y + p.th[10] * z + p.th[11];   // row 2 on D
    float ix = ((gx + 1.f) / 2.f) * (float)(p.Wi - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(p.Hi - 1);
    float iz = ((gz + 1.f) / 2.f) * (float)(p.Di - 1);
    // ---- image: trilinear, zeros outside --------------------------------------------------------
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    float wgt[8];
    int64_t off[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int xx = x0 + (c & 1), yy = y0 + ((c >> 1) & 1), zz = z0 + (c >> 2);
      float wv = ((c & 1) ? tx : 1.f - tx) * (((c >> 1) & 1) ? ty : 1.f - ty) * ((c >> 2) ? tz : 1.f - tz);
      bool ok = xx >= 0 && xx < p.Wi && yy >= 0 && yy < p.Hi && zz >= 0 && zz < p.Di;
      wgt[c] = ok ? wv : 0.f;
      off[c] = ok ? ((int64_t)zz * p.Hi + yy) * p.Wi + xx : 0;
    }
    for (int ch = 0; ch < p.C; ++ch) {
      const float* src = p.img + (int64_t)ch * Sin;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) a += wgt[c] * src[off[c]];
      p.oimg[(int64_t)ch * total + i] = a;
    }
    // ---- label: nearest (round half to even), zeros outside ----------------------------------------
    if (p.lab) {
      int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy), zn = (int)nearbyintf(iz);
      int64_t v = 0;
      if (xn >= 0 && xn < p.Wi && yn >= 0 && yn < p.Hi && zn >= 0 && zn < p.Di) {
        int64_t o = ((int64_t)zn * p.Hi + yn) * p.Wi + xn;
        v = p.lab_bytes == 1 ? (int64_t)((const int8_t*)p.lab)[o] : ((const int64_t*)p.lab)[o];
      }
      p.olab[i] = v;
    }
  }
}

__global__ void __launch_bounds__(NT) k_crop3d(const float* img, const void* lab, float* oimg, void* olab, int C,
                                               int Di, int Hi, int Wi, int Do, int Ho, int Wo, int d0, int h0,
                                               int w0, int lab_bytes) {
  const int64_t total = (int64_t)Do * Ho * Wo, Sin = (int64_t)Di * Hi * Wi;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int w = (int)(i % Wo);
    int64_t r = i / Wo;
    int h = (int)(r % Ho), d = (int)(r / Ho);
    int64_t o = ((int64_t)(d0 + d) * Hi + (h0 + h)) * Wi + (w0 + w);
    for (int ch = 0; ch < C; ++ch) oimg[(int64_t)ch * total + i] = img[(int64_t)ch * Sin + o];
    if (lab) {
      if (lab_bytes == 1) ((int8_t*)olab)[i] = ((const int8_t*)lab)[o];
      else ((int64_t*)olab)[i] = ((const int64_t*)lab)[o];
    }
  }
}

// per-channel partial (min, max, n, mean, M2); grid = (blocks, C)
__global__ void __launch_bounds__(NT) k_chan_stats_partial(const float* __restrict__ x, int64_t S,
                                                           float* __restrict__ partials) {
  const float* xc = x + (int64_t)blockIdx.y * S;
  float mn = INFINITY, mx = -INFINITY, cnt = 0.f, sh = 0.f, s0 = 0.f, s1 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < S; i += (int64_t)gridDim.x * NT) {
    float v = xc[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
    if (cnt == 0.f) sh = v;
    float d = v - sh;
    s0 += d; s1 += d * d; cnt += 1.f;
  }
  __shared__ float red[NT][5];
  Moments m = moments_from_shifted(cnt, sh, s0, s1);
  red[threadIdx.x][0] = mn; red[threadIdx.x][1] = mx; red[threadIdx.x][2] = m.n; red[threadIdx.x][3] = m.mean;
  red[threadIdx.x][4] = m.m2;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      float* a = red[threadIdx.x];
      const float* b = red[threadIdx.x + s];
      a[0] = fminf(a[0], b[0]); a[1] = fmaxf(a[1], b[1]);
      Moments ma = {a[2], a[3], a[4]}, mb = {b[2], b[3], b[4]};
      ma = moments_merge(ma, mb);
      a[2] = ma.n; a[3] = ma.mean; a[4] = ma.m2;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;
    for (int j = 0; j < 5; ++j) o[j] = red[0][j];
  }
}
// out[c] = (min, max, mean, unbiased std)
__global__ void k_chan_stats_final(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const int c = blockIdx.x;
  double mn = INFINITY, mx = -INFINITY, n = 0.0, mean = 0.0, M2 = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const float* q = partials + ((size_t)c * nblk + b) * 5;
    mn = fmin(mn, (double)q[0]); mx = fmax(mx, (double)q[1]);
    double nb = q[2];
    if (nb > 0.0) {
      double nn = n + nb, d = (double)q[3] - mean;
      mean += d * (nb / nn);
      M2 += (double)q[4] + d * d * (n * nb / nn);
      n = nn;
    }
  }
  out[c * 4 + 0] = (float)mn; out[c * 4 + 1] = (float)mx; out[c * 4 + 2] = (float)mean;
  out[c * 4 + 3] = (float)sqrt(n > 1.0 ? M2 / (n - 1.0) : 0.0);   // torch.std: unbiased
}

// mode 0: y = x*a + b                       (brightness_multiply / additive; prm = {a, b} per channel)
// mode 1: y = pow((x-min)/rng, g)*rng + min (gamma, :126;  prm = {g}; st = stats of x)
// mode 2: y = (x - m2)/s2*s1 + m1           (retain_stats, :128-130; st = stats of gamma output, st2 = of input)
// mode 3: y = clamp((x-mean)*f + mean, min, max) (contrast, :158-161; prm = {f, 1 = preserve_range False: no clamp})
// mode 4: y = x + noise*a + b               (gaussian_noise, :17; prm = {std, mean})
// stats index: stat_c = per_channel ? c : 0;  prm index likewise (prm_stride floats per channel)
__global__ void __launch_bounds__(NT) k_intensity(const float* __restrict__ x, float* __restrict__ y, int C,
                                                  int64_t S, int mode, const float* __restrict__ prm,
                                                  int prm_per_channel, const float* __restrict__ st,
                                                  const float* __restrict__ st2, int st_per_channel,
                                                  const float* __restrict__ noise) {
  const int c = blockIdx.y;
  const float* pr = prm + (prm_per_channel ? c : 0) * 2;
  const float* s1p = st ? st + (st_per_channel ? c : 0) * 4 : nullptr;
  const float* s2p = st2 ? st2 + (st_per_channel ? c : 0) * 4 : nullptr;
  const float* xc = x + (int64_t)c * S;
  float* yc = y + (int64_t)c * S;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < S; i += (int64_t)gridDim.x * NT) {
    float v = xc[i], r;
    if (mode == 0) r = v * pr[0] + pr[1];
    else if (mode == 1) { float mn = s1p[0], rng = s1p[1] - s1p[0]; r = powf((v - mn) / rng, pr[0]) * rng + mn; }
    else if (mode == 2) r = (v - s1p[2]) / s1p[3] * s2p[3] + s2p[2];
    else if (mode == 3) { float m = s1p[2]; r = (v - m) * pr[0] + m; if (pr[1] == 0.f) r = fminf(fmaxf(r, s1p[0]), s1p[1]); }
    else r = v + noise[(int64_t)c * S + i] * pr[0] + pr[1];
    yc[i] = r;
  }
}

// 1-D zero-padded correlation along one axis: len = extent of the axis, stride = element stride of it,
// `inner`/`outer` enumerate the other two axes.  taps <= 31.
struct BlurParams { const float* x; float* y; int64_t n_lines; int len; int64_t stride; int64_t inner; int64_t outer_stride; int k; float g[31]; };
__global__ void __launch_bounds__(NT) k_blur_axis(BlurParams p) {
  const int64_t total = p.n_lines * p.len;
  const int r = p.k / 2;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    // decode so that consecutive threads touch consecutive memory: innermost memory index first
    int64_t line, pos;
    if (p.stride == 1) { pos = i % p.len; line = i / p.len; }
    else { int64_t in = i % p.inner; int64_t rest = i / p.inner; pos = rest % p.len; line = (rest / p.len) * p.inner + in; }
    int64_t base = (line / p.inner) * p.outer_stride + (line % p.inner);
    float a = 0.f;
    for (int t = 0; t < p.k; ++t) {
      int64_t q = pos + t - r;
      if (q >= 0 && q < p.len) a += p.g[t] * p.x[base + q * p.stride];
    }
    p.y[base + pos * p.stride] = a;
  }
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cbim

using namespace cbim;

static int launch_status(const char* what) {
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "%s launch: %s", what, hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_affine_sample3d(const float* img, const void* lab, int lab_bytes, const float* theta12,
                                    float* out_img, int64_t* out_lab, int C, int Di, int Hi, int Wi, int Do,
                                    int Ho, int Wo, int od0, int oh0, int ow0, void* stream) {
  CBIM_CHECK(img && out_img && theta12, CBIM_EINVAL, "null argument");
  CBIM_CHECK(!lab || (out_lab && (lab_bytes == 1 || lab_bytes == 8)), CBIM_EUNSUPPORTED, "label must be int8 or int64");
  CBIM_CHECK(od0 >= 0 && oh0 >= 0 && ow0 >= 0 && od0 + Do <= Di && oh0 + Ho <= Hi && ow0 + Wo <= Wi, CBIM_EINVAL,
             "crop window outside the sampling grid");
  AffineParams p;
  p.img = img; p.lab = lab; p.oimg = out_img; p.olab = out_lab; p.C = C; p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.Do = Do; p.Ho = Ho; p.Wo = Wo; p.od0 = od0; p.oh0 = oh0; p.ow0 = ow0; p.lab_bytes = lab_bytes;
  for (int i = 0; i < 12; ++i) p.th[i] = theta12[i];   // HOST pointer: 12 floats drawn by the host RNG
  CBIM_LAUNCH(k_affine_sample3d, dim3(grid_for((int64_t)Do * Ho * Wo)), dim3(NT), 0, (hipStream_t)stream, p);
  return launch_status("affine_sample3d");
}

extern "C" int cbim_crop3d(const float* img, const void* lab, int lab_bytes, float* out_img, void* out_lab, int C,
                           int Di, int Hi, int Wi, int Do, int Ho, int Wo, int d0, int h0, int w0, void* stream) {
  CBIM_CHECK(img && out_img, CBIM_EINVAL, "null argument");
  CBIM_CHECK(!lab || (out_lab && (lab_bytes == 1 || lab_bytes == 8)), CBIM_EUNSUPPORTED, "label must be int8 or int64");
  CBIM_CHECK(d0 >= 0 && h0 >= 0 && w0 >= 0 && d0 + Do <= Di && h0 + Ho <= Hi && w0 + Wo <= Wi, CBIM_EINVAL,
             "crop window outside the volume");
  CBIM_LAUNCH(k_crop3d, dim3(grid_for((int64_t)Do * Ho * Wo)), dim3(NT), 0, (hipStream_t)stream, img, lab, out_img,
              out_lab, C, Di, Hi, Wi, Do, Ho, Wo, d0, h0, w0, lab_bytes);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

static int stat_blocks(int64_t S) {
  int64_t b = (S + NT * 16 - 1) / (NT * 16);
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" size_t cbim_chan_stats_workspace(int C, int64_t S) { return (size_t)C * stat_blocks(S) * 5 * sizeof(float); }

// stats: float [C][4] = (min, max, mean, unbiased std) over the S elements of each channel
extern "C" int cbim_chan_stats(const float* x, int C, int64_t S, float* stats, void* workspace, size_t ws_bytes,
                               void* stream) {
  CBIM_CHECK(x && stats && workspace && ws_bytes >= cbim_chan_stats_workspace(C, S), CBIM_EWORKSPACE, "bad workspace");
  int nb = stat_blocks(S);
  hipStream_t st = (hipStream_t)stream;
  CBIM_LAUNCH(k_chan_stats_partial, dim3(nb, C), dim3(NT), 0, st, x, S, (float*)workspace);
  CBIM_LAUNCH(k_chan_stats_final, dim3(C), dim3(64), 0, st, (const float*)workspace, nb, stats);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_intensity(const float* x, float* y, int C, int64_t S, int mode, const float* prm,
                              int prm_per_channel, const float* st, const float* st2, int st_per_channel,
                              const float* noise, void* stream) {
  CBIM_CHECK(x && y && prm && mode >= 0 && mode <= 4, CBIM_EINVAL, "bad argument");
  CBIM_CHECK((mode != 1 && mode != 3) || st, CBIM_EINVAL, "mode needs statistics");
  CBIM_CHECK(mode != 2 || (st && st2), CBIM_EINVAL, "mode 2 needs both statistics");
  CBIM_CHECK(mode != 4 || noise, CBIM_EINVAL, "mode 4 needs noise");
  CBIM_LAUNCH(k_intensity, dim3(grid_for(S), C), dim3(NT), 0, (hipStream_t)stream, x, y, C, S, mode, prm,
              prm_per_channel, st, st2, st_per_channel, noise);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// y = blur of x ([C][D][H][W]) with the normalised 1-D kernel g[k] along D, H and W; tmp: one more volume
extern "C" int cbim_gaussian_blur3d(const float* x, float* y, float* tmp, int C, int D, int H, int W,
                                    const float* g_host, int k, void* stream) {
  CBIM_CHECK(x && y && tmp && g_host, CBIM_EINVAL, "null argument");
  CBIM_CHECK(k >= 1 && k <= 31 && (k & 1), CBIM_EUNSUPPORTED, "blur kernel size %d unsupported", k);
  hipStream_t st = (hipStream_t)stream;
  BlurParams p;
  p.k = k;
  for (int i = 0; i < k; ++i) p.g[i] = g_host[i];
  const int64_t S = (int64_t)D * H * W;
  const int64_t total = (int64_t)C * S;
  // W axis: x -> y
  p.x = x; p.y = y; p.len = W; p.stride = 1; p.inner = 1; p.outer_stride = W; p.n_lines = (int64_t)C * D * H;
  CBIM_LAUNCH(k_blur_axis, dim3(grid_for(total)), dim3(NT), 0, st, p);
  // H axis: y -> tmp   (lines: (c,d) outer with stride H*W, w inner)
  p.x = y; p.y = tmp; p.len = H; p.stride = W; p.inner = W; p.outer_stride = (int64_t)H * W; p.n_lines = (int64_t)C * D * W;
  CBIM_LAUNCH(k_blur_axis, dim3(grid_for(total)), dim3(NT), 0, st, p);
  // D axis: tmp -> y   (lines: c outer with stride D*H*W, (h,w) inner)
  p.x = tmp; p.y = y; p.len = D; p.stride = (int64_t)H * W; p.inner = (int64_t)H * W; p.outer_stride = S; p.n_lines = (int64_t)C * H * W;
  CBIM_LAUNCH(k_blur_axis, dim3(grid_for(total)), dim3(NT), 0, st, p);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(augment)
