// Video input pipeline on the device: decoded uint8 mouth clips -> the model-ready batch.
//   reference, per sample on dataloader workers: datasets.py:187-196 (uint8 -> float /255, Grayscale, NormalizeVideo), the config's video_transform
//   (AV cfg:82-89: RandomCrop 88x88, RandomHorizontalFlip, TimeMaskSecond with the clip mean, transforms.py:108-126), align_video_to_audio
//   (transforms.py:169-180: zero frames left/right), then CollateFn zero-pads to the batch maximum (collate_fn.py:143-146).
// Here: the random decisions (crop origin, flip, mask intervals) are drawn on the host in the reference's call order and passed in `geom`/`masks`;
// the pixel work is three HBM-bound launches over the whole batch:
//   1. video_crop_norm_kernel   one workgroup per output frame: crop + flip + (grayscale) + normalise, zero frames for alignment / batch padding, per-frame sums
//   2. video_mask_values_kernel one workgroup per clip: the sequential "fill with the mean of the clip as masked so far" recurrence on the frame sums
//   3. video_mask_fill_kernel   one workgroup per masked frame: constant fill
#include "common.h"
#include "avec_hip.h"

#define VG_TV 0       // geom[b][8]: frames in the decoded clip
#define VG_H 1        //            source height
#define VG_W 2        //            source width
#define VG_CY 3       //            crop origin
#define VG_CX 4
#define VG_FLIP 5
#define VG_PADL 6     //            zero frames in front (alignment)
#define VG_NMASK 7

// Pixel arithmetic of the reference chain: ConvertImageDtype (u / 255), Grayscale (0.2989 r + 0.587 g + 0.114 b), NormalizeVideo ((x - mean) / std).
// hipcc contracts a*b+c into an fma (also through the __f*_rn spellings), which changes the last bit; so the products come from a host-made table
// lut[c][u] = w_c * (u / 255) (w = 1 for gray clips) and the kernel only adds, subtracts and divides -- nothing left to fuse.
__device__ __forceinline__ float px_value(const unsigned char* p, int C, const float* lut, float mean, float stdv) {
  const float v = C == 1 ? lut[p[0]] : (lut[p[0]] + lut[256 + p[1]]) + lut[512 + p[2]];
  return (v - mean) / stdv;
}

__global__ __launch_bounds__(256) void video_crop_norm_kernel(const unsigned char* __restrict__ clips, const long long* __restrict__ clip_off, const int* __restrict__ geom,
                                                              int C, const float* __restrict__ lut_g, float mean, float stdv, float* __restrict__ out, float* __restrict__ frame_sum, int Tout, int OH, int OW) {
  __shared__ float lut[768];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int* g = geom + b * 8;
  const int ts = t - g[VG_PADL];
  float* dst = out + ((long long)b * Tout + t) * OH * OW;
  const int n4 = OH * OW / 4;
  float acc = 0.f;
  if (ts < 0 || ts >= g[VG_TV]) {
    for (int q = tid; q < n4; q += 256) *(float4*)(dst + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int i = tid; i < C * 256; i += 256) lut[i] = lut_g[i];
    __syncthreads();
    const int W = g[VG_W], cy = g[VG_CY], cx = g[VG_CX], flip = g[VG_FLIP];
    const unsigned char* src = clips + clip_off[b] + (long long)ts * g[VG_H] * W * C;
    const int qw = OW / 4;
    for (int q = tid; q < n4; q += 256) {
      const int y = q / qw, x0 = (q - y * qw) * 4;
      const unsigned char* row = src + ((long long)(cy + y) * W + cx) * C;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int x = x0 + e; v[e] = px_value(row + (flip ? OW - 1 - x : x) * C, C, lut, mean, stdv); }
      *(float4*)(dst + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
      acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  if (frame_sum) {
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) frame_sum[(long long)b * Tout + t] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// fill[b][t] = NaN (frame kept) or the value the frame is overwritten with.  Mask k takes the mean of the clip as left by masks 0..k-1 (transforms.py:121-124).
__global__ __launch_bounds__(256) void video_mask_values_kernel(const int* __restrict__ geom, const int* __restrict__ masks, int max_masks, const float* __restrict__ frame_sum,
                                                                float* __restrict__ fill, int Tout, int frame_px, int mean_frame) {
  extern __shared__ double fsum[];                 // [Tout] sums of the clip's frames, then a 4-entry reduction pad
  double* red = fsum + Tout;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* g = geom + b * 8;
  const int tv = g[VG_TV], padl = g[VG_PADL], nm = g[VG_NMASK];
  for (int t = tid; t < Tout; t += 256) { fill[(long long)b * Tout + t] = __builtin_nanf(""); if (t < tv) fsum[t] = (double)frame_sum[(long long)b * Tout + padl + t]; }
  __syncthreads();
  for (int k = 0; k < nm; ++k) {
    const int s = masks[((long long)b * max_masks + k) * 2], e = masks[((long long)b * max_masks + k) * 2 + 1];
    float m = 0.f;
    if (mean_frame) {
      double part = 0.0;
      for (int t = tid; t < tv; t += 256) part += fsum[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if ((tid & 63) == 0) red[tid >> 6] = part;
      __syncthreads();
      m = (float)(((red[0] + red[1]) + (red[2] + red[3])) / ((double)tv * frame_px));
      __syncthreads();
    }
    for (int t = s + tid; t < e && t < tv; t += 256) { fsum[t] = (double)m * frame_px; fill[(long long)b * Tout + padl + t] = m; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void video_mask_fill_kernel(const float* __restrict__ fill, float* __restrict__ out, int Tout, int frame_px) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float m = fill[(long long)b * Tout + t];
  if (m != m) return;
  float* dst = out + ((long long)b * Tout + t) * frame_px;
  for (int q = threadIdx.x; q < frame_px / 4; q += 256) *(float4*)(dst + q * 4) = make_float4(m, m, m, m);
}

extern "C" int avec_video_input(const unsigned char* clips, const long long* clip_off, const int* geom, const int* masks, int max_masks, int channels, const float* lut, float mean, float stdv,
                                int mean_frame, float* out, float* frame_ws, int B, int Tout, int OH, int OW, hipStream_t st) {
  AVEC_CHECK_ARG(clips && clip_off && geom && out && lut, "video_input: null buffer");
  AVEC_CHECK_ARG(channels == 1 || channels == 3, "video_input: 1 (gray) or 3 (RGB) channels, got %d", channels);
  AVEC_CHECK_ARG(OW % 4 == 0 && B > 0 && Tout > 0 && OH > 0, "video_input: crop width must be a multiple of 4 (got %d), B/T/H > 0", OW);
  AVEC_CHECK_ARG(max_masks == 0 || (masks && frame_ws), "video_input: masks need the mask list and a [2][B][Tout] float workspace");
  AVEC_CHECK_ARG(Tout <= 4096 && B <= 65535, "video_input: at most 4096 frames per clip and 65535 clips per call");
  float* frame_sum = max_masks ? frame_ws : nullptr;
  hipLaunchKernelGGL(video_crop_norm_kernel, dim3(Tout, B), dim3(256), 0, st, clips, clip_off, geom, channels, lut, mean, stdv, out, frame_sum, Tout, OH, OW);
  AVEC_LAUNCH_CHECK();
  if (max_masks) {
    float* fill = frame_ws + (long long)B * Tout;
    hipLaunchKernelGGL(video_mask_values_kernel, dim3(B), dim3(256), (Tout + 4) * sizeof(double), st, geom, masks, max_masks, frame_sum, fill, Tout, OH * OW, mean_frame);
    AVEC_LAUNCH_CHECK();
    hipLaunchKernelGGL(video_mask_fill_kernel, dim3(Tout, B), dim3(256), 0, st, fill, out, Tout, OH * OW);
    AVEC_LAUNCH_CHECK();
  }
  return 0;
}
