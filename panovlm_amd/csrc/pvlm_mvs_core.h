// Per-texel bodies of the panoramic MVS scoring kernel (photometric term of ScorePixel, mvs/MVS.cpp:774-923, with
// FillPixelPatch :637-680 and the bilinear Sample).  Plain float arithmetic in the reference's order; host/device so
// that tests/cpp/mvs_math_check.cpp can drive the same functions serially on a machine without a GPU.
#pragma once
#include "pvlm_exact_math.h"
#include <cfloat>
#include <cmath>

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__
#endif

namespace pvlm_mvs {

// exp / sin / cos / acos of a FLOAT (the reference calls the std:: float overloads; their last bit depends on the libm
// version): the correctly rounded float result, taken as the double function rounded to float — the same definition the
// test oracle uses, so that device, host-compiled check and oracle agree bit for bit.
// On the device each is ONE out-of-line function: inlined, the double-precision routines (1-4 KB each) were repeated at every call
// site and k_mvs_propagate_flow_spec<1> came to 70 KB of code — more than the 64 KB instruction cache two CUs share.
#if defined(__HIP_DEVICE_COMPILE__)
#define PVLM_MVS_MATH __attribute__((noinline))
#else
#define PVLM_MVS_MATH
#endif
PVLM_HD inline PVLM_MVS_MATH float f_exp(float x) { return (float)exp((double)x); }
PVLM_HD inline float f_exp_inline(float x) { return (float)exp((double)x); }   // FillPixelPatch: once per texel, one call site per kernel
PVLM_HD inline PVLM_MVS_MATH float f_sin(float x) { return (float)sin((double)x); }
PVLM_HD inline PVLM_MVS_MATH float f_cos(float x) { return (float)cos((double)x); }
PVLM_HD inline PVLM_MVS_MATH float f_acos(float x) { return (float)acos((double)x); }

// FastAtan2<float> (base/Math.h:15-29): polynomial evaluated in double (double literals), rounded to float on assignment
PVLM_HD inline float fast_atan2f(float y, float x) {
  const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
  const float mn = ay < ax ? ay : ax, mx = ax < ay ? ay : ax;
  const float a = mn / (mx + (float)DBL_EPSILON);
  const float s = a * a;
  float r = ((-0.04432655554792128 * s + 0.1555786518463281) * s - 0.3258083974640975) * s * a + 0.9997878412794807 * a;
  if (ay > ax) r = 1.57079632679489661923 - r;
  if (x < 0) r = 3.14159265358979323846 - r;
  if (y < 0) r = -r;
  return r;
}

// Equirectangular::CamToImage<float> (sensors/Equirectangular.h:50-51, :84-85)
PVLM_HD inline void cam_to_image(int rows, int cols, const float* X, float* px) {
  // the reference's statements are (float)sqrt((double)(..)), lon / (2.0 * pi), lat / pi in double: same floats, fewer
  // instructions (pvlm_exact_math.h; the oracle keeps the statements as written)
  const float lon = fast_atan2f(X[0], X[2]);
  const float lat = -fast_atan2f(X[1], pvlm_exact::sqrt_via_double(X[0] * X[0] + X[2] * X[2]));
  px[0] = (float)(cols * (0.5 + pvlm_exact::div_two_pi(lon)));
  px[1] = (float)(rows * (0.5 - pvlm_exact::div_pi(lat)));
}

// Equirectangular::ImageToCam<float>(pixel, 1.f) of an integer pixel — the PreComputeI2C table (Equirectangular.cpp:12-19)
PVLM_HD inline void unit_ray(int rows, int cols, int col, int row, float* cam) {
  const float sx = (float)((2 * (float)col / cols - 1) * 3.14159265358979323846);
  const float sy = (float)((0.5 - (float)row / rows) * 3.14159265358979323846);
  const float cy = (float)cos((double)sy);
  cam[0] = 1.f * cy * (float)sin((double)sx);
  cam[1] = -1.f * (float)sin((double)sy);
  cam[2] = 1.f * cy * (float)cos((double)sx);
}

// BGR2HSV (util/Visualization.cpp:57-77) scaled as MVS::DepthImageToCloud does (mvs/MVS.cpp:2095-2099), and its "sky blue" box test
PVLM_HD inline bool sky_colour(const unsigned char* bgr) {
  const float r = bgr[2] / 255.f, g = bgr[1] / 255.f, b = bgr[0] / 255.f;
  const float c_max = fmaxf(r, fmaxf(g, b)), c_min = fminf(r, fminf(g, b));
  float h = 0.f, s = 0.f, v = 0.f;
  if (c_max != 0.f) {
    const float delta = c_max - c_min;
    if (c_max == r) h = 60.f * ((g - b) / delta + (float)(6 * (g < b)));
    else if (c_max == g) h = 60.f * ((b - r) / delta + 2.f);
    else h = 60.f * ((r - g) / delta + 4.f);
    h = h / 360.f; s = delta / c_max; v = c_max;
  }
  h *= 180.f; s *= 255.f; v *= 255.f;
  return h >= 100.f && h <= 124.f && s >= 43.f && s <= 200.f && v >= 150.f && v <= 255.f;
}
// the pixel test of MVS::DepthImageToCloud / DepthNormalToCloud (mvs/MVS.cpp:2083-2085, :2120-2121, :2099)
PVLM_HD inline bool cloud_keeps(float depth, float max_depth, const unsigned char* bgr, bool filter_sky) {
  if (depth <= 0 || (double)depth >= (double)max_depth * 0.8) return false;
  return !(filter_sky && sky_colour(bgr));
}
// TranslatePoint<float, double> (base/Geometry.hpp:545-551) of unit ray x depth; R_wc n for the normal (mvs/MVS.cpp:2133-2138)
PVLM_HD inline void cloud_point(const float* ray, float depth, const double* T_wc, float* xyz) {
  const float pc[3] = {ray[0] * depth, ray[1] * depth, ray[2] * depth};
  for (int k = 0; k < 3; ++k) xyz[k] = (float)(pc[0] * T_wc[4 * k] + pc[1] * T_wc[4 * k + 1] + pc[2] * T_wc[4 * k + 2] + T_wc[4 * k + 3]);
}
PVLM_HD inline void cloud_normal(const float* n, const double* T_wc, float* out) {
  for (int k = 0; k < 3; ++k) out[k] = (float)(T_wc[4 * k] * (double)n[0] + T_wc[4 * k + 1] * (double)n[1] + T_wc[4 * k + 2] * (double)n[2]);
}

PVLM_HD inline int num_texels(int half_window, int step) {
  const int w = 2 * half_window + 1, q = w / step + (step > 1 ? 1 : 0);
  return q * q;
}
// texel k of the window -> offsets (di, dj) from the top-left corner, as the reference's i / j loops visit them
PVLM_HD inline void texel_offset(int half_window, int step, int k, int* di, int* dj) {
  const int w = 2 * half_window + 1, q = w / step + (step > 1 ? 1 : 0);
  *di = (k / q) * step; *dj = (k % q) * step;
}

// FillPixelPatch, per texel: un-normalised bilateral weight and the grey value
PVLM_HD inline void patch_texel(const unsigned char* gray, int cols, int px, int py, int half_window, int step, int k, float* weight, float* texel) {
  int di, dj; texel_offset(half_window, step, k, &di, &dj);
  const int row = py - half_window + di, col = px - half_window + dj;
  const float sigma_color = (float)(-1.f / (2 * 0.2 * 0.2));
  const float sigma_spatial = -1.f / (2.f * half_window * half_window);
  const unsigned char center = gray[(size_t)py * cols + px], tex = gray[(size_t)row * cols + col];
  float wColor = (tex - center) / 255.f;
  wColor = wColor * wColor * sigma_color;
  const float wSpatial = ((float)((col - px) * (col - px)) + (float)((row - py) * (row - py))) * sigma_spatial;
  *weight = f_exp_inline(wColor + wSpatial);
  *texel = tex;
}

// H = R_nr + (1.f / d) * t_nr * normal^T   (row-major 3x3)
PVLM_HD inline void homography(const float* R, const float* t, const float* normal, float d, float* H) {
  const float inv_d = 1.f / d;
  for (int i = 0; i < 3; ++i) { const float ti = inv_d * t[i]; for (int j = 0; j < 3; ++j) H[3 * i + j] = R[3 * i + j] + ti * normal[j]; }
}

// The four grey values of a bilinear tap at pixel offset `at` (its top-left pixel).  From the image itself: four byte loads.  From its
// 2x2-QUAD copy (QuadImage: one 32-bit word per pixel = I(x,y) | I(x+1,y) << 8 | I(x,y+1) << 16 | I(x+1,y+1) << 24, built once per neighbour
// image on the device — 4 bytes per pixel, 66 MB for a 5.7K view): ONE dword load, the bytes unpacked by v_cvt_f32_ubyte0..3.  The taps of
// the 49 texels of every scoring are the gathers the MVS kernels wait on; the same four values either way.
struct Tap4 { unsigned char p00, p01, p10, p11; };
struct QuadImage { const unsigned* q; };
PVLM_HD inline Tap4 tap4(const unsigned char* gray, int cols, size_t at) { const unsigned char* p = gray + at; return Tap4{p[0], p[1], p[cols], p[cols + 1]}; }
PVLM_HD inline Tap4 tap4(QuadImage img, int cols, size_t at) {
  (void)cols;
  const unsigned v = img.q[at];
  return Tap4{(unsigned char)(v & 255u), (unsigned char)((v >> 8) & 255u), (unsigned char)((v >> 16) & 255u), (unsigned char)(v >> 24)};
}

// one texel of the neighbour patch: project the reference texel's unit ray through H, test frame.IsInside(x1, 1, 1),
// sample bilinearly.  Returns false when the projection leaves the image (the whole neighbour is then skipped).
// the same with the reference texel's unit ray already in hand (the wave scorer loads it once per scoring, not once per neighbour image)
template <class Img>
PVLM_HD inline bool neighbour_texel_ray(const float* uv, Img nei_gray, int rows, int cols, const float* H, float* value) {
  float X1[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += H[3 * r + c] * uv[c]; X1[r] = s; }
  float x1[2];
  cam_to_image(rows, cols, X1, x1);
  if (!(x1[0] >= 1 && x1[1] >= 1 && x1[0] < cols - 1 && x1[1] < rows - 1)) return false;
  const int lx = (int)x1[0], ly = (int)x1[1];
  const float fx = x1[0] - lx, fy = x1[1] - ly, ax = 1.f - fx, ay = 1.f - fy;
  const Tap4 p = tap4(nei_gray, cols, (size_t)ly * cols + lx);
  *value = (p.p00 * ax + p.p01 * fx) * ay + (p.p10 * ax + p.p11 * fx) * fy;
  return true;
}
PVLM_HD inline bool neighbour_texel(const float* unit, const unsigned char* nei_gray, int rows, int cols, const float* H, int px, int py, int half_window,
                                    int step, int k, float* value) {
  int di, dj; texel_offset(half_window, step, k, &di, &dj);
  const float* uv = unit + 3 * ((size_t)(py - half_window + di) * cols + (px - half_window + dj));
  float X1[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += H[3 * r + c] * uv[c]; X1[r] = s; }
  float x1[2];
  cam_to_image(rows, cols, X1, x1);
  if (!(x1[0] >= 1 && x1[1] >= 1 && x1[0] < cols - 1 && x1[1] < rows - 1)) return false;
  const int lx = (int)x1[0], ly = (int)x1[1];
  const float fx = x1[0] - lx, fy = x1[1] - ly, ax = 1.f - fx, ay = 1.f - fy;
  const unsigned char* p = nei_gray + (size_t)ly * cols + lx;
  *value = (p[0] * ax + p[1] * fx) * ay + (p[cols] * ax + p[cols + 1] * fx) * fy;
  return true;
}

// Equirectangular::ImageToCam<float>(pixel, r) (sensors/Equirectangular.h:102-103, :125-128, :149-152)
PVLM_HD inline void image_to_cam(int rows, int cols, const float* px, float r, float* cam) {
  const float sx = (float)((2 * px[0] / cols - 1) * 3.14159265358979323846);
  const float sy = (float)((0.5 - px[1] / rows) * 3.14159265358979323846);
  const float cy = f_cos(sy);
  cam[0] = r * cy * f_sin(sx);
  cam[1] = -r * f_sin(sy);
  cam[2] = r * cy * f_cos(sx);
}

// Sample(img, pt, functor) of mvs/MVS.cpp:1445-1467 with the functor of ScorePixel :873: |depth0 - d| / depth0 < 0.03f
PVLM_HD inline float sample_depth(const float* img, int cols, float x, float y, float depth0) {
  const int lx = (int)x, ly = (int)y;
  const float fx = x - lx, fy = y - ly, x1 = 1.f - fx, y1 = 1.f - fy;
  const float x0y0 = img[(size_t)ly * cols + lx], x1y0 = img[(size_t)ly * cols + lx + 1];
  const float x0y1 = img[(size_t)(ly + 1) * cols + lx], x1y1 = img[(size_t)(ly + 1) * cols + lx + 1];
  const bool b00 = fabsf(depth0 - x0y0) / depth0 < 0.03f, b10 = fabsf(depth0 - x1y0) / depth0 < 0.03f;
  const bool b01 = fabsf(depth0 - x0y1) / depth0 < 0.03f, b11 = fabsf(depth0 - x1y1) / depth0 < 0.03f;
  if (!b00 && !b10 && !b01 && !b11) return INFINITY;
  return (float)(y1 * (x1 * (b00 ? x0y0 : (b10 ? x1y0 : (b01 ? x0y1 : x1y1))) + fx * (b10 ? x1y0 : (b00 ? x0y0 : (b11 ? x1y1 : x0y1)))) +
                 fy * (x1 * (b01 ? x0y1 : (b11 ? x1y1 : (b00 ? x0y0 : x1y0))) + fx * (b11 ? x1y1 : (b01 ? x0y1 : (b10 ? x1y0 : x0y0)))));
}

// geometric-consistency adjustment of one neighbour's score (ScorePixel :857-893)
PVLM_HD inline float geometric_adjust(float score, int rows, int cols, const float* X0, const float* R, const float* t, const float* nei_depth) {
  const float geometric_weight = 0.2;
  float consistency = 2;
  float X1[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R[3 * r + c] * X0[c]; X1[r] = s + t[r]; }
  const float depth0 = (float)sqrt((double)X1[0] * X1[0] + (double)X1[1] * X1[1] + (double)X1[2] * X1[2]);
  float x1[2];
  cam_to_image(rows, cols, X1, x1);
  score = 1 - score;
  if (x1[0] >= 1 && x1[1] >= 1 && x1[0] < cols - 1 && x1[1] < rows - 1) {
    const float depth1 = sample_depth(nei_depth, cols, x1[0], x1[1], depth0);
    if (depth1 != INFINITY && depth1 != -INFINITY) {   // !isinf(depth1)
      float cam[3], t_rn[3], Xb[3];
      image_to_cam(rows, cols, x1, depth1, cam);
      for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += (-R[3 * c + r]) * t[c]; t_rn[r] = s; }
      for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R[3 * c + r] * cam[c]; Xb[r] = s + t_rn[r]; }
      float cosang = X0[0] * Xb[0] + X0[1] * Xb[1] + X0[2] * Xb[2];
      const float n1 = sqrtf(X0[0] * X0[0] + X0[1] * X0[1] + X0[2] * X0[2]), n2 = sqrtf(Xb[0] * Xb[0] + Xb[1] * Xb[1] + Xb[2] * Xb[2]);
      cosang /= (n1 * n2);
      const float ang = cosang >= 1.f ? 0.f : (cosang <= -1.f ? (float)3.14159265358979323846 : f_acos(cosang));
      const float diff_angle = (float)(ang * 180.0 / 3.14159265358979323846);
      consistency = diff_angle < consistency ? diff_angle : consistency;
    }
  }
  score += geometric_weight * consistency;
  score = 1 - score;
  return fminf(1.f, fmaxf(-1.f, score));
}

// ---- depth-map fusion filter: ProjectDepthConfToRef (mvs/MVS.cpp:2011-2070, depth only) + FilterDepthImage (:1735-1790) ----
#ifndef PVLM_ATOMIC_MIN_U32
#define PVLM_ATOMIC_MIN_U32(ptr, v) atomicMin((ptr), (v))
#endif

PVLM_HD inline unsigned float_bits(float f) { union { float f; unsigned u; } c; c.f = f; return c.u; }
PVLM_HD inline float bits_float(unsigned u) { union { float f; unsigned u; } c; c.u = u; return c.f; }

// R_rn = R_nr^T, t_rn = (-R_rn) t_nr
PVLM_HD inline void inverse_pose(const float* R_nr, const float* t_nr, float* R_rn, float* t_rn) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_rn[3 * r + c] = R_nr[3 * c + r];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += (-R_rn[3 * r + c]) * t_nr[c]; t_rn[r] = s; }
}

// one neighbour pixel e: move it to the reference camera and splat its range onto the four integer pixels around the
// projection; every target keeps the smallest range (the reference's "if (d != 0 && d < range) continue; d = range").
// proj_bits: rows x cols float bit patterns initialised to +inf (0x7f800000) — ranges are >= 0, so unsigned order = float order.
PVLM_HD inline void project_splat(int rows, int cols, const float* unit, const float* nei_depth, const float* R_rn, const float* t_rn, long long e,
                                  unsigned* proj_bits) {
  const float dn = nei_depth[e];
  const float pn[3] = {unit[3 * e] * dn, unit[3 * e + 1] * dn, unit[3 * e + 2] * dn};
  float pr[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R_rn[3 * r + c] * pn[c]; pr[r] = s + t_rn[r]; }
  const float range = (float)sqrt((double)pr[0] * pr[0] + (double)pr[1] * pr[1] + (double)pr[2] * pr[2]);
  float px[2];
  cam_to_image(rows, cols, pr, px);
  const int xs[2] = {(int)ceilf(px[0]), (int)floorf(px[0])}, ys[2] = {(int)ceilf(px[1]), (int)floorf(px[1])};
  const unsigned bits = float_bits(range);
  for (int b = 0; b < 2; ++b)
    for (int a = 0; a < 2; ++a) {
      const int x = xs[a], y = ys[b];
      if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
      PVLM_ATOMIC_MIN_U32(&proj_bits[(size_t)y * cols + x], bits);
    }
}
PVLM_HD inline float projected_depth(const unsigned* proj_bits, size_t e) { const unsigned b = proj_bits[e]; return b == 0x7f800000u ? 0.f : bits_float(b); }

// FilterDepthImage for one reference pixel; proj_bits: n_neighbors images one after the other
PVLM_HD inline void filter_pixel(int rows, int cols, int n_neighbors, const unsigned* proj_bits, const float* depth, const float* conf,
                                 const unsigned char* depth_constant, float thr, long long e, float* depth_filter, float* conf_filter) {
  const size_t npix = (size_t)rows * cols;
  depth_filter[e] = 0.f;
  if (conf_filter) conf_filter[e] = 0.f;
  const float d = depth[e];
  if (d <= 0) return;
  const float loose = thr * 1.2f, strict = thr * 0.8f;
  const int row = (int)(e / cols), col = (int)(e % cols);
  int similar = 0;
  for (int b = 0; b < n_neighbors; ++b) { const float dn = projected_depth(proj_bits + npix * b, e); if (dn > 0 && fabsf((d - dn) / d) < strict) similar++; }
  if (similar < 2) return;
  similar = 0;
  const int ox[4] = {-1, 1, 0, 0}, oy[4] = {0, 0, 1, -1};
  for (int b = 0; b < n_neighbors; ++b)
    for (int k = 0; k < 4; ++k) {
      const int x = col + ox[k], y = row + oy[k];
      if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
      const float dn = projected_depth(proj_bits + npix * b, (size_t)y * cols + x);
      if (dn > 0 && fabsf((d - dn) / d) < loose) similar++;
    }
  const bool keep_constant = depth_constant && depth_constant[e];
  if (similar < 5 && !keep_constant) return;
  depth_filter[e] = d;
  if (conf && conf_filter) conf_filter[e] = conf[e];
}

// ---- FilterDepthImageRefine (mvs/MVS.cpp:1794-1890) with ProjectDepthConfToRef projecting depth AND confidence ----
// Sequentially (raster order) a target pixel ends with the smallest range it received and the confidence of the LAST
// source pixel whose range equals that minimum (a write happens whenever !(d != 0 && d < range)).  Order-free form:
// one 64-bit atomicMin of key = range_bits << 32 | (0xffffffff - source index); rows x cols keys initialised to ~0.
#ifndef PVLM_ATOMIC_MIN_U64
#define PVLM_ATOMIC_MIN_U64(ptr, v) atomicMin((ptr), (v))
#endif
PVLM_HD inline void project_splat_conf(int rows, int cols, const float* unit, const float* nei_depth, const float* R_rn, const float* t_rn, long long e,
                                       unsigned long long* proj_key) {
  const float dn = nei_depth[e];
  const float pn[3] = {unit[3 * e] * dn, unit[3 * e + 1] * dn, unit[3 * e + 2] * dn};
  float pr[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R_rn[3 * r + c] * pn[c]; pr[r] = s + t_rn[r]; }
  const float range = (float)sqrt((double)pr[0] * pr[0] + (double)pr[1] * pr[1] + (double)pr[2] * pr[2]);
  float px[2];
  cam_to_image(rows, cols, pr, px);
  const int xs[2] = {(int)ceilf(px[0]), (int)floorf(px[0])}, ys[2] = {(int)ceilf(px[1]), (int)floorf(px[1])};
  const unsigned long long key = ((unsigned long long)float_bits(range) << 32) | (unsigned long long)(0xffffffffu - (unsigned)e);
  for (int b = 0; b < 2; ++b)
    for (int a = 0; a < 2; ++a) {
      const int x = xs[a], y = ys[b];
      if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
      PVLM_ATOMIC_MIN_U64(&proj_key[(size_t)y * cols + x], key);
    }
}
PVLM_HD inline void projected_depth_conf(const unsigned long long* proj_key, const float* nei_conf, size_t e, float* d, float* c) {
  const unsigned long long k = proj_key[e];
  if (k == ~0ull) { *d = 0.f; *c = 0.f; return; }
  *d = bits_float((unsigned)(k >> 32));
  *c = nei_conf[0xffffffffu - (unsigned)(k & 0xffffffffull)];
}

struct RefineViews { const float* conf[16]; float R[16][9]; float t[16][3]; int n; };   // neighbour conf_map + T_nr

// one reference pixel; proj_key: n images one after the other.  conf is in-out (zeroed where depth <= 0).
PVLM_HD inline void refine_pixel(int rows, int cols, const RefineViews& nv, const unsigned long long* proj_key, const float* unit, const float* depth,
                                 float* conf, const unsigned char* depth_constant, float thr, float min_depth, float max_depth, long long e,
                                 float* depth_filter, float* conf_filter) {
  const size_t npix = (size_t)rows * cols;
  depth_filter[e] = 0.f;
  conf_filter[e] = 0.f;
  const float d = depth[e];
  if (d <= 0) { conf[e] = 0.f; return; }
  const float loose = thr * 1.2f;
  float positive = conf[e], negative = 0.f, avg = d * positive;
  int n_pos = 0, n_neg = 0;
  bool bad = false;
  for (int n = nv.n - 1; n >= 0; --n) {
    float dn, cn;
    projected_depth_conf(proj_key + npix * n, nv.conf[n], (size_t)e, &dn, &cn);
    if (dn <= 0 && n_pos + n_neg + n < 2) { bad = true; break; }
    if (fabsf((d - dn) / d) < loose) {
      avg += dn * cn;
      positive += cn;
      n_pos += 1;
    } else {
      if (dn < d) negative += cn;           // occlusion
      else {                                // free-space violation
        const float X0[3] = {unit[3 * e] * d, unit[3 * e + 1] * d, unit[3 * e + 2] * d};
        float X1[3], x1[2];
        for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += nv.R[n][3 * r + c] * X0[c]; X1[r] = s + nv.t[n][r]; }
        cam_to_image(rows, cols, X1, x1);
        const int xr = (int)roundf(x1[0]), yr = (int)roundf(x1[1]);
        if (xr >= 0 && yr >= 0 && xr < cols && yr < rows) {
          const float c = nv.conf[n][(size_t)yr * cols + xr];
          negative += (c > 0 ? c : cn);
        } else negative += cn;
      }
      n_neg += 1;
    }
  }
  if (!bad) {
    avg /= positive;
    if (n_pos >= 2 && positive > negative && avg >= min_depth && avg <= max_depth) {
      depth_filter[e] = avg;
      conf_filter[e] = positive - negative;
      return;
    }
  }
  if (depth_constant && depth_constant[e]) { depth_filter[e] = d; conf_filter[e] = 1.f; }
}


// ---- PatchMatch sweep: PropagateCheckerBoard (mvs/MVS.cpp:1098-1129) -> ProcessPixel (:721-772) -> PerturbDepthNormal3
// (:1254-1320) with InterpolatePixel (:1923-1935), CorrectNormal (:1953-1971), PerturbNormal / PerturbDepth /
// GenerateRandomNormal (:1368-1431) and the smoothness term of ScorePixel (:843-857).
// Random draws: upstream shares one time-seeded cv::RNG between the threads of an `omp parallel for` (a data race, no
// run repeats).  Here draw k of pixel e in pass p is a hash of (seed, p, e, k); float conversion as cv::RNG
// (next() * 2^-32; uniform(a, b) = that * (b - a) + a).
PVLM_HD inline unsigned random_u32(unsigned long long seed, unsigned long long pixel, unsigned k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (pixel + 1) + 0xD1B54A32D192ED03ull * (unsigned long long)(k + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32);
}
PVLM_HD inline unsigned long long pass_seed(unsigned long long seed, int pass) {
  return seed * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull * (unsigned long long)(pass + 1);
}
struct Rng {
  unsigned long long seed, pixel; unsigned k;
  PVLM_HD float next01() { return (float)random_u32(seed, pixel, k++) * 2.3283064365386962890625e-10f; }
  PVLM_HD float uniform(float a, float b) { return next01() * (b - a) + a; }
};

PVLM_HD inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

PVLM_HD inline float interpolate_pixel(const float* view_ray, const float* unit_n, float depth, const float* normal, float min_depth, float max_depth) {
  const float X1[3] = {unit_n[0] * depth, unit_n[1] * depth, unit_n[2] * depth};
  const float dnorm = dot3(view_ray, normal);
  if (fabsf(dnorm) < 1e-6) return depth;
  const float depth_new = dot3(X1, normal) / dnorm;
  if (depth_new >= min_depth && depth_new <= max_depth) return depth_new;
  return depth;
}

// rotates a normal that looks away from the camera back to just past 90 degrees; Eigen::AngleAxisf(rad, axis) with the
// un-normalised axis upstream passes (AngleAxis::toRotationMatrix written out)
PVLM_HD inline void correct_normal(const float* viewDir, float* normal) {
  const float cosAngLen = dot3(normal, viewDir);
  if (cosAngLen >= 0) {
    const float axis[3] = {normal[1] * viewDir[2] - normal[2] * viewDir[1], normal[2] * viewDir[0] - normal[0] * viewDir[2], normal[0] * viewDir[1] - normal[1] * viewDir[0]};
    const float v = (f_acos(cosAngLen) - (float)1.57079632679489661923) * 1.01f;
    const float rad = (-0.001f < v) ? -0.001f : v;                       // std::min(v, -0.001f)
    const float sn = f_sin(rad), c = f_cos(rad);
    const float sin_axis[3] = {sn * axis[0], sn * axis[1], sn * axis[2]};
    const float cos1_axis[3] = {(1.f - c) * axis[0], (1.f - c) * axis[1], (1.f - c) * axis[2]};
    float R[9];
    float tmp = cos1_axis[0] * axis[1];
    R[1] = tmp - sin_axis[2]; R[3] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * axis[2];
    R[2] = tmp + sin_axis[1]; R[6] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * axis[2];
    R[5] = tmp - sin_axis[0]; R[7] = tmp + sin_axis[0];
    R[0] = cos1_axis[0] * axis[0] + c; R[4] = cos1_axis[1] * axis[1] + c; R[8] = cos1_axis[2] * axis[2] + c;
    const float x = R[0] * normal[0] + R[1] * normal[1] + R[2] * normal[2];
    const float y = R[3] * normal[0] + R[4] * normal[1] + R[5] * normal[2];
    const float z = R[6] * normal[0] + R[7] * normal[1] + R[8] * normal[2];
    normal[0] = x; normal[1] = y; normal[2] = z;
  }
}

// The two pieces of wave-uniform transcendental arithmetic of a hypothesis — the smoothness factors of the (up to four) close
// neighbours and the three sine / cosine pairs of a normal perturbation — go through the scorer object: on the host they are
// the loops below; on the GPU, where a wave-uniform value costs a full wave instruction however few lanes need it, the wave
// evaluates the four factors (resp. the three angles) in four (three) LANES at once and broadcasts them (WaveScorer in
// pvlm_mvs.hip) — the same function on the same arguments, a quarter (third) of the instructions.
struct SerialMath {
  PVLM_HD void sincos3(const float* a, float* sn, float* cs) const { for (int k = 0; k < 3; ++k) { sn[k] = f_sin(a[k]); cs[k] = f_cos(a[k]); } }
};

template <class Math>
PVLM_HD inline void perturb_normal(const Math& math, Rng& rng, const float* normal, float perturbation, float* out) {
  const float a1 = (rng.next01() - 0.5f) * perturbation;
  const float a2 = (rng.next01() - 0.5f) * perturbation;
  const float a3 = (rng.next01() - 0.5f) * perturbation;
  const float ang[3] = {a1, a2, a3};
  float sn3[3], cs3[3];
  math.sincos3(ang, sn3, cs3);
  const float sin_a1 = sn3[0], sin_a2 = sn3[1], sin_a3 = sn3[2];
  const float cos_a1 = cs3[0], cos_a2 = cs3[1], cos_a3 = cs3[2];
  float R[9];
  R[0] = cos_a2 * cos_a3;
  R[1] = -cos_a2 * sin_a3;
  R[2] = sin_a2;
  R[3] = cos_a1 * sin_a3 + cos_a3 * sin_a1 * sin_a2;
  R[4] = cos_a1 * cos_a3 - sin_a1 * sin_a2 * sin_a3;
  R[5] = -cos_a2 * sin_a1;
  R[6] = sin_a1 * sin_a3 - cos_a1 * cos_a3 * sin_a2;
  R[7] = cos_a3 * sin_a1 + cos_a1 * sin_a2 * sin_a3;
  R[8] = cos_a1 * cos_a2;
  out[0] = R[0] * normal[0] + R[1] * normal[1] + R[2] * normal[2];
  out[1] = R[3] * normal[0] + R[4] * normal[1] + R[5] * normal[2];
  out[2] = R[6] * normal[0] + R[7] * normal[1] + R[8] * normal[2];
}
PVLM_HD inline float perturb_depth(Rng& rng, float depth, float perturbation) {
  const float max_depth = (1 + perturbation) * depth, min_depth = (1 - perturbation) * depth;
  return rng.uniform(0.f, 1.f) * (max_depth - min_depth) + min_depth;
}
PVLM_HD inline void generate_random_normal(Rng& rng, const float* view_ray, float* normal) {
  float v1 = 0.0f, v2 = 0.0f, s = 2.0f;
  while (s >= 1.0f) {
    v1 = 2.0f * rng.uniform(0.f, 1.f) - 1.0f;
    v2 = 2.0f * rng.uniform(0.f, 1.f) - 1.0f;
    s = v1 * v1 + v2 * v2;
  }
  const float s_norm = sqrtf(1.0f - s);
  normal[0] = 2.0f * v1 * s_norm; normal[1] = 2.0f * v2 * s_norm; normal[2] = 1.0f - 2.0f * s;
  if (dot3(normal, view_ray) > 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
}

struct ClosePixel { float point[3]; float normal[3]; float depth; };   // NeighborPixel, mvs/MVS.h:59-64

// one close neighbour's factor of the smoothness term: (1 - bonusDepth * exp(dd^2 sigmaD)) * (1 - bonusNormal * exp(da^2 sigmaN))
PVLM_HD inline float smooth_factor(const float* plane, const ClosePixel& c, const float* normal, float depth) {
  const float smoothBonus = 0.95f, smoothBonusDepth = 1.f - smoothBonus, smoothBonusNormal = (float)((1.f - smoothBonus) * 0.96);
  const float smoothSigmaDepth = -1.f / (2.f * 0.02f * 0.02f), smoothSigmaNormal = -1.f / (2.f * 0.22f * 0.22f);
  const float diff_distance = fabsf(plane[0] * c.point[0] + plane[1] * c.point[1] + plane[2] * c.point[2] + plane[3]) / depth;
  const float factorDepth = f_exp(diff_distance * diff_distance * smoothSigmaDepth);
  const float cosang = normal[0] * c.normal[0] + normal[1] * c.normal[1] + normal[2] * c.normal[2];
  const float diff_angle = cosang >= 1.f ? 0.f : (cosang <= -1.f ? (float)3.14159265358979323846 : f_acos(cosang));
  const float factorNormal = f_exp(diff_angle * diff_angle * smoothSigmaNormal);
  return (1.f - smoothBonusDepth * factorDepth) * (1.f - smoothBonusNormal * factorNormal);
}
struct SerialFactors {
  PVLM_HD void smooth_factors(const float* plane, const ClosePixel* close, int n_close, const float* normal, float depth, float* factors) const {
    for (int c = 0; c < n_close; ++c) factors[c] = smooth_factor(plane, close[c], normal, depth);
  }
};
// applied to one neighbour image's clamped NCC (ScorePixel :843-857)
PVLM_HD inline float smooth_score(float score, const float* factors, int n_close) {
  if (n_close <= 0) return score;
  score = 1 - score;
  for (int q = 0; q < n_close; ++q) score *= factors[q];
  score = 1 - score;
  return fminf(1.f, fmaxf(-1.f, score));
}
// the same with compile-time indices (n_close <= 4): for callers that hold the factors in registers (the wave-per-pixel scorer; the
// thread-per-pixel kernels are at their register budget and keep the loop above)
PVLM_HD inline float smooth_score_static(float score, const float* factors, int n_close) {
  if (n_close <= 0) return score;
  score = 1 - score;
#pragma unroll
  for (int q = 0; q < 4; ++q) if (q < n_close) score *= factors[q];
  score = 1 - score;
  return fminf(1.f, fmaxf(-1.f, score));
}

struct SweepArgs {
  int rows, cols;
  const float* unit;                       // PreComputeI2C table
  const float* depth; const float* normal; // the maps (read: the other colour of the checkerboard)
  const unsigned char* depth_constant;     // may be null
  float min_depth, max_depth;
};

// ProcessPixel + PerturbDepthNormal3 for pixel (px, py).  score(normal, depth, factors, n_close) -> aggregated ScorePixel
// value; factors = the smooth_factor of every close neighbour for that hypothesis (n_close = 0: no smoothness term).
// depth / normal / conf: the pixel's own state, updated in place by the caller-visible references.
// n_prop / pdx / pdy: the pixels whose hypotheses are propagated, as the reference's sweeps pass them to ProcessPixel — the
// four direct neighbours for the checkerboard (:1113-1114), (left, up) or (right, down) for the sequential sweep (:1073, :1091).
template <class Scorer>
PVLM_HD inline void process_pixel(const SweepArgs& A, Rng& rng, int px, int py, Scorer& score, float& depth, float* normal, float& conf,
                                  int n_prop, const int* pdx, const int* pdy) {
  const int rows = A.rows, cols = A.cols;
  const size_t e = (size_t)py * cols + px;
  const bool keep_depth_constant = A.depth_constant && A.depth_constant[e];
  const float* view_ray = A.unit + 3 * e;
  ClosePixel close[4] = {}; int n_close = 0;
  {
    const int cx[4] = {px - 1, px, px, px + 1}, cy[4] = {py, py - 1, py + 1, py};
    for (int q = 0; q < 4; ++q) {
      if (!(cx[q] >= 0 && cy[q] >= 0 && cx[q] < cols && cy[q] < rows)) continue;
      const size_t ne = (size_t)cy[q] * cols + cx[q];
      const float d = A.depth[ne];
      if (d <= 0) continue;
      ClosePixel& c = close[n_close++];
      for (int k = 0; k < 3; ++k) { c.point[k] = A.unit[3 * ne + k] * d; c.normal[k] = A.normal[3 * ne + k]; }
      c.depth = d;
    }
  }
  float factors[4];
  // ---- propagation from the four direct neighbours (:749-771)
  {
    for (int q = 0; q < n_prop; ++q) {
      const int nxq = px + pdx[q], nyq = py + pdy[q];
      if (!(nxq >= 0 && nyq >= 0 && nxq < cols && nyq < rows)) continue;
      const size_t ne = (size_t)nyq * cols + nxq;
      float depth_neighbor = A.depth[ne];
      if (depth_neighbor <= 0) continue;
      float normal_neighbor[3] = {A.normal[3 * ne], A.normal[3 * ne + 1], A.normal[3 * ne + 2]};
      depth_neighbor = keep_depth_constant ? depth : interpolate_pixel(view_ray, A.unit + 3 * ne, depth_neighbor, normal_neighbor, A.min_depth, A.max_depth);
      correct_normal(view_ray, normal_neighbor);
      const float X0[3] = {view_ray[0] * depth_neighbor, view_ray[1] * depth_neighbor, view_ray[2] * depth_neighbor};
      const float plane[4] = {normal_neighbor[0], normal_neighbor[1], normal_neighbor[2], -dot3(normal_neighbor, X0)};
      score.smooth_factors(plane, close, n_close, normal_neighbor, depth_neighbor, factors);
      const float newconf = score(normal_neighbor, depth_neighbor, factors, n_close);
      if (conf < newconf) { conf = newconf; depth = depth_neighbor; normal[0] = normal_neighbor[0]; normal[1] = normal_neighbor[1]; normal[2] = normal_neighbor[2]; }
    }
  }
  // ---- PerturbDepthNormal3 (perturb_depth = !keep_depth_constant, perturb_normal = true)
  const bool perturb = !keep_depth_constant;
  const float scaleRanges[12] = {1.f, 0.5f, 0.25f, 0.125f, 0.0625f, 0.03125f, 0.015625f, 0.0078125f, 0.00390625f, 0.001953125f, 0.0009765625f, 0.00048828125f};
  const float thConfSmall = (float)(0.55 * 0.2f), thConfBig = (float)(0.55 * 0.4f), thConfRand = (float)(0.55 * 0.9f);
  unsigned idxScaleRange = 0;
  if (1 - conf <= thConfSmall) idxScaleRange = 2;
  else if (1 - conf <= thConfBig) idxScaleRange = 1;
  else if (1 - conf >= thConfRand) {
    bool refine = false;
    for (int iter = 0; iter < 6; iter++) {
      const float depth_random = perturb ? rng.uniform(A.min_depth, A.max_depth) : depth;
      float normal_random[3];
      generate_random_normal(rng, view_ray, normal_random);
      const float nconf = score(normal_random, depth_random, factors, 0);
      if (nconf > conf) {
        conf = nconf; depth = depth_random; normal[0] = normal_random[0]; normal[1] = normal_random[1]; normal[2] = normal_random[2];
        if (1 - nconf < thConfRand) { refine = true; break; }
      }
    }
    if (!refine) return;
  }
  float scaleRange = scaleRanges[idxScaleRange];
  const float depthRange = (float)(depth * 0.02);
  const float angleRange = (float)(30.f / 180.f * 3.14159265358979323846);
  for (int iter = 0; iter < 6; iter++) {
    const float depth_perturb = perturb ? perturb_depth(rng, depth, scaleRange * depthRange) : depth;
    float normal_perturb[3];
    perturb_normal(score, rng, normal, scaleRange * angleRange, normal_perturb);
    if (dot3(normal_perturb, view_ray) >= 0) continue;
    const float X0[3] = {view_ray[0] * depth_perturb, view_ray[1] * depth_perturb, view_ray[2] * depth_perturb};
    const float plane[4] = {normal_perturb[0], normal_perturb[1], normal_perturb[2], -dot3(normal_perturb, X0)};
    score.smooth_factors(plane, close, n_close, normal_perturb, depth_perturb, factors);
    const float nconf = score(normal_perturb, depth_perturb, factors, n_close);
    if (nconf > conf) {
      conf = nconf; depth = depth_perturb; normal[0] = normal_perturb[0]; normal[1] = normal_perturb[1]; normal[2] = normal_perturb[2];
      idxScaleRange++;
      scaleRange = scaleRanges[idxScaleRange];
    }
  }
}

// ---- ProcessPixel with the independent hypotheses of a pixel scored side by side ------------------------------------------------
// The score of a hypothesis does not depend on the pixel's current confidence, only the DECISIONS do.  So
//   * the propagated hypotheses (the walk's two already-updated neighbours) can be scored together and then accepted or not one
//     after the other, exactly as the sequential loop would;
//   * the six refinements of PerturbDepthNormal3 chain only through acceptances: iteration i perturbs the state left by the last
//     ACCEPTED iteration before it, with draws whose position in the pixel's counter-based random stream is fixed (4 draws per
//     iteration, 3 when the depth is kept).  A batch of W consecutive iterations is therefore built from the current state as if
//     none of them were accepted, scored together, and resolved in order: everything up to and including the first accepted one
//     is what the sequential loop would have done; the rest of the batch is discarded and the next batch starts behind it.
// Same results as process_pixel, bit for bit (tests/test_mvs_cpu.py drives both through the serial scorer); the chain of dependent
// scorings drops from 2 + 6 to 1 + (6 / expected batch progress).  On the GPU a batch is the W waves of a workgroup
// (k_mvs_propagate_diag_spec: a single view's anti-diagonal is at most min(rows, cols) pixels — a sixth of the chip's wave slots with
// one wave per pixel).  The random phase (1 - conf >= thConfRand, rejection-sampled normals: a variable number of draws) stays
// a chain; every executor of the batch runs it redundantly.
struct Hypothesis { float normal[3]; float depth; float conf; int valid; };
#ifndef PVLM_MVS_SPEC_STAT
#define PVLM_MVS_SPEC_STAT(i) do { } while (0)   // measured variant of pvlm_mvs.hip (-DPVLM_MVS_FLOW_CLOCK=1): outcomes per pixel
#endif

// hypothesis -> its score (plane through the point, smoothness factors of the close neighbours, ScorePixel)
template <class Scorer>
PVLM_HD inline float score_hypothesis(Scorer& score, const float* view_ray, const ClosePixel* close, int n_close, const Hypothesis& h) {
  const float X0[3] = {view_ray[0] * h.depth, view_ray[1] * h.depth, view_ray[2] * h.depth};
  const float plane[4] = {h.normal[0], h.normal[1], h.normal[2], -dot3(h.normal, X0)};
  float factors[4];
  score.smooth_factors(plane, close, n_close, h.normal, h.depth, factors);
  return score(h.normal, h.depth, factors, n_close);
}

// Batch runner of the host check: the W "waves" one after the other.  build(w, h) fills hypothesis w of the batch (valid = 0: not scored).
template <class Scorer>
struct SerialBatch {
  Scorer* score; int W;
  PVLM_HD int width() const { return W; }
  template <class Build>
  PVLM_HD void run(int n, const float* view_ray, const ClosePixel* close, int n_close, Build&& build, Hypothesis* out) {
    for (int w = 0; w < n; ++w) {
      build(w, out[w]);
      out[w].conf = out[w].valid ? score_hypothesis(*score, view_ray, close, n_close, out[w]) : -1.f;
    }
  }
  PVLM_HD Scorer& single() { return *score; }
};

// State of the four direct neighbours of a pixel — slot 0 (x - 1, y), 1 (x, y - 1), 2 (x, y + 1), 3 (x + 1, y), the order ProcessPixel
// collects its close pixels in — as the sweep shows it to this pixel.  gather_around reads it from the maps; the data-flow kernels
// (pvlm_mvs.hip, K13r) receive the two slots the walk has just updated from the workgroup that computed them instead.
struct Around { float depth[4]; float normal[4][3]; int inside[4]; };
PVLM_HD inline int around_slot(int dx, int dy) { return dx < 0 ? 0 : (dy < 0 ? 1 : (dy > 0 ? 2 : 3)); }
PVLM_HD inline void gather_around(const SweepArgs& A, int px, int py, Around& ar) {
  const int cx[4] = {px - 1, px, px, px + 1}, cy[4] = {py, py - 1, py + 1, py};
  for (int q = 0; q < 4; ++q) {
    ar.inside[q] = cx[q] >= 0 && cy[q] >= 0 && cx[q] < A.cols && cy[q] < A.rows;
    ar.depth[q] = 0.f; ar.normal[q][0] = ar.normal[q][1] = ar.normal[q][2] = 0.f;
    if (!ar.inside[q]) continue;
    const size_t ne = (size_t)cy[q] * A.cols + cx[q];
    ar.depth[q] = A.depth[ne];
    for (int k = 0; k < 3; ++k) ar.normal[q][k] = A.normal[3 * ne + k];
  }
}

// The close pixels of ProcessPixel (:735-747) from an Around: the direct neighbours that hold a hypothesis, compacted in slot order.
// Written with compile-time indices only (the s-th entry is selected, not addressed): an array indexed by a running count would live
// in scratch memory on the GPU, and every scoring reads all of it.
PVLM_HD inline int build_close(const SweepArgs& A, int px, int py, const Around& ar, ClosePixel* close) {
  const int cols = A.cols;
  const int cx[4] = {px - 1, px, px, px + 1}, cy[4] = {py, py - 1, py + 1, py};
  int n_close = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) { close[s].depth = 0.f; for (int k = 0; k < 3; ++k) { close[s].point[k] = 0.f; close[s].normal[k] = 0.f; } }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float d = ar.depth[q];
    const bool take = ar.inside[q] && d > 0;
    const size_t ne = take ? (size_t)cy[q] * cols + cx[q] : (size_t)py * cols + px;
    ClosePixel c;
    for (int k = 0; k < 3; ++k) { c.point[k] = A.unit[3 * ne + k] * d; c.normal[k] = ar.normal[q][k]; }
    c.depth = d;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (take && s == n_close) close[s] = c;
    n_close += take ? 1 : 0;
  }
  return n_close;
}

template <class Batch>
PVLM_HD inline void process_pixel_close(const SweepArgs& A, Rng& rng, int px, int py, Batch& batch, const Around& ar, const ClosePixel* close, int n_close,
                                        float& depth, float* normal, float& conf, int n_prop, const int* pdx, const int* pdy) {
  const int cols = A.cols;
  const size_t e = (size_t)py * cols + px;
  const bool keep_depth_constant = A.depth_constant && A.depth_constant[e];
  const float* view_ray = A.unit + 3 * e;
  const int W = batch.width();
  Hypothesis hyp[4];
  // ---- propagation (:749-771): the hypotheses do not depend on each other's outcome
  for (int base = 0; base < n_prop; base += W) {
    const int n = n_prop - base < W ? n_prop - base : W;
    const float depth_now = depth;
    batch.run(n, view_ray, close, n_close, [&](int w, Hypothesis& h) {
      h.valid = 0;
      const int q = base + w;
      const int slot = around_slot(pdx[q], pdy[q]);
      if (!ar.inside[slot]) return;
      const size_t ne = (size_t)(py + pdy[q]) * cols + (px + pdx[q]);
      float depth_neighbor = ar.depth[slot];
      if (depth_neighbor <= 0) return;
      float normal_neighbor[3] = {ar.normal[slot][0], ar.normal[slot][1], ar.normal[slot][2]};
      // keep_depth_constant: the pixel's own depth, which no acceptance changes (an accepted neighbour hands over that same depth)
      depth_neighbor = keep_depth_constant ? depth_now : interpolate_pixel(view_ray, A.unit + 3 * ne, depth_neighbor, normal_neighbor, A.min_depth, A.max_depth);
      correct_normal(view_ray, normal_neighbor);
      h.normal[0] = normal_neighbor[0]; h.normal[1] = normal_neighbor[1]; h.normal[2] = normal_neighbor[2]; h.depth = depth_neighbor; h.valid = 1;
    }, hyp);
#pragma unroll
    for (int w = 0; w < 4; ++w)                              // w < n <= 4: compile-time indices keep hyp[] in registers
      if (w < n && hyp[w].valid && conf < hyp[w].conf) { conf = hyp[w].conf; depth = hyp[w].depth; normal[0] = hyp[w].normal[0]; normal[1] = hyp[w].normal[1]; normal[2] = hyp[w].normal[2]; PVLM_MVS_SPEC_STAT(0); }
  }
  // ---- PerturbDepthNormal3
  const bool perturb = !keep_depth_constant;
  const float scaleRanges[12] = {1.f, 0.5f, 0.25f, 0.125f, 0.0625f, 0.03125f, 0.015625f, 0.0078125f, 0.00390625f, 0.001953125f, 0.0009765625f, 0.00048828125f};
  const float thConfSmall = (float)(0.55 * 0.2f), thConfBig = (float)(0.55 * 0.4f), thConfRand = (float)(0.55 * 0.9f);
  unsigned idxScaleRange = 0;
  if (1 - conf <= thConfSmall) idxScaleRange = 2;
  else if (1 - conf <= thConfBig) idxScaleRange = 1;
  else if (1 - conf >= thConfRand) {
    PVLM_MVS_SPEC_STAT(1);
    bool refine = false;
    float factors[4];
    for (int iter = 0; iter < 6; iter++) {
      const float depth_random = perturb ? rng.uniform(A.min_depth, A.max_depth) : depth;
      float normal_random[3];
      generate_random_normal(rng, view_ray, normal_random);
      const float nconf = batch.single()(normal_random, depth_random, factors, 0);
      if (nconf > conf) {
        conf = nconf; depth = depth_random; normal[0] = normal_random[0]; normal[1] = normal_random[1]; normal[2] = normal_random[2];
        if (1 - nconf < thConfRand) { refine = true; break; }
      }
    }
    if (!refine) return;
  }
  const float angleRange = (float)(30.f / 180.f * 3.14159265358979323846);
  const unsigned draws = perturb ? 4u : 3u;                 // per refinement iteration: perturb_depth (1, when the depth moves) + perturb_normal (3)
  // the depth range is taken once, before the loop, from the depth at that point — like upstream (:1291) — not per batch
  const float depthRange = (float)(depth * 0.02);
  for (int base = 0; base < 6;) {
    const int n = 6 - base < W ? 6 - base : W;
    const float scaleRange = scaleRanges[idxScaleRange];
    const float depth_now = depth;
    const float normal_now[3] = {normal[0], normal[1], normal[2]};
    const unsigned k0 = rng.k;
    batch.run(n, view_ray, close, n_close, [&](int w, Hypothesis& h) {
      Rng r = rng; r.k = k0 + draws * (unsigned)w;          // iteration base + w, as if base .. base + w - 1 were all rejected
      h.depth = perturb ? perturb_depth(r, depth_now, scaleRange * depthRange) : depth_now;
      perturb_normal(batch.single(), r, normal_now, scaleRange * angleRange, h.normal);
      h.valid = dot3(h.normal, view_ray) >= 0 ? 0 : 1;     // `continue` (:1303-1304): the draws are spent, nothing is scored
    }, hyp);
    int taken = n;
    bool accepted = false;
#pragma unroll
    for (int w = 0; w < 4; ++w)                              // the first accepted one ends the batch
      if (!accepted && w < n && hyp[w].valid && hyp[w].conf > conf) {
        conf = hyp[w].conf; depth = hyp[w].depth; normal[0] = hyp[w].normal[0]; normal[1] = hyp[w].normal[1]; normal[2] = hyp[w].normal[2];
        idxScaleRange++;
        taken = w + 1;
        accepted = true;
        PVLM_MVS_SPEC_STAT(3);
      }
    rng.k = k0 + draws * (unsigned)taken;
    base += taken;
  }
}

template <class Batch>
PVLM_HD inline void process_pixel_around(const SweepArgs& A, Rng& rng, int px, int py, Batch& batch, const Around& ar, float& depth, float* normal, float& conf,
                                         int n_prop, const int* pdx, const int* pdy) {
  ClosePixel close[4];
  const int n_close = build_close(A, px, py, ar, close);
  process_pixel_close(A, rng, px, py, batch, ar, close, n_close, depth, normal, conf, n_prop, pdx, pdy);
}

template <class Batch>
PVLM_HD inline void process_pixel_spec(const SweepArgs& A, Rng& rng, int px, int py, Batch& batch, float& depth, float* normal, float& conf,
                                       int n_prop, const int* pdx, const int* pdy) {
  Around ar;
  gather_around(A, px, py, ar);
  process_pixel_around(A, rng, px, py, batch, ar, depth, normal, conf, n_prop, pdx, pdy);
}

// ---- one pixel per THREAD ------------------------------------------------------------------------------------------------------
// FillPixelPatch + ScorePixel as the reference runs them: one pixel, its window texel after texel, the sums in index order.  The
// per-pixel arrays are COLUMNS of tables shared by many pixels — element k of a column at [k * stride]:
//   w   the normalised bilateral weights: a table in global memory (k_mvs_propagate_lane: [texel][pixel of the pass], written by the
//       thread that reads it, coalesced), because it is the one array that lives for the whole pixel;
//   t1  the current neighbour image's texels: the workgroup's LDS ([texel][lane], 12.5 KB per wave at 7 x 7);
//   the weighted centred reference texel (t0 - mean) * w is recomputed from the grey byte, the mean and w: the same two operations
//   FillPixelPatch performs (:668-672), so the same float.
// On the GPU this is 64 pixels per wave: no lane idles on a 49-texel window, no wave-uniform value is computed 64 times, no
// cross-lane sums.  On the host (tests/cpp/mvs_math_check.cpp) the strides are 1.
struct ColumnPatch { float* w; size_t w_stride; float* t1; int t1_stride; float mean, sq0; bool inside; };

PVLM_HD inline float ref_texel(const unsigned char* gray, int cols, int px, int py, int half_window, int step, int k) {
  int di, dj; texel_offset(half_window, step, k, &di, &dj);
  return (float)gray[(size_t)(py - half_window + di) * cols + (px - half_window + dj)];
}

// SHARED: several threads fill the SAME column with the same values (the four threads of a pixel in k_mvs_propagate_diag_batch_quad).
// The one-owner form parks the raw weight in the column and normalises it in place — a read-modify-write that four unsynchronised
// threads may not do on one address; the shared form sums the weights first and stores only the final value (stores of equal values:
// idempotent), at the price of evaluating the 49 weights twice.  Same arithmetic, same bits.
template <bool SHARED = false>
PVLM_HD inline void fill_patch_column(const unsigned char* ref_gray, int rows, int cols, int px, int py, int half_window, int step, int n, ColumnPatch& P) {
  P.sq0 = 0.f; P.mean = 0.f;
  P.inside = px >= half_window && py >= half_window && px < cols - half_window && py < rows - half_window;
  if (!P.inside) return;
  const size_t st = P.w_stride;
  float wsum = 0.f;
  float mean = 0.f;
  if (SHARED) {
    for (int k = 0; k < n; ++k) { float w, t; patch_texel(ref_gray, cols, px, py, half_window, step, k, &w, &t); wsum += w; }                             // :659
    for (int k = 0; k < n; ++k) { float w, t; patch_texel(ref_gray, cols, px, py, half_window, step, k, &w, &t); w = w / wsum; P.w[k * st] = w; mean += w * ref_texel(ref_gray, cols, px, py, half_window, step, k); }
  } else {
  for (int k = 0; k < n; ++k) { float w, t; patch_texel(ref_gray, cols, px, py, half_window, step, k, &w, &t); P.w[k * st] = w; wsum += w; }            // :659
  for (int k = 0; k < n; ++k) { const float w = P.w[k * st] / wsum; P.w[k * st] = w; mean += w * ref_texel(ref_gray, cols, px, py, half_window, step, k); }   // :662-664
  }
  float sq0 = 0.f;
  for (int k = 0; k < n; ++k) { const float t = ref_texel(ref_gray, cols, px, py, half_window, step, k) - mean; const float tmp = t * P.w[k * st]; sq0 += t * tmp; }   // :668-672
  P.mean = mean; P.sq0 = sq0;
}

// neighbour_texel in two halves, so that the four byte loads of texel k + 1 can be in flight while texel k is consumed:
// the projection (where the tap lands, the bilinear fractions) and the interpolation of the four grey values.
struct TexelTap { size_t at; float fx, fy; bool ok; };
PVLM_HD inline TexelTap texel_tap(const float* uv, int rows, int cols, const float* H) {
  float X1[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += H[3 * r + c] * uv[c]; X1[r] = s; }
  float x1[2];
  cam_to_image(rows, cols, X1, x1);
  TexelTap t;
  t.ok = x1[0] >= 1 && x1[1] >= 1 && x1[0] < cols - 1 && x1[1] < rows - 1;                  // frame.IsInside(x1, 1, 1)
  const int lx = t.ok ? (int)x1[0] : 0, ly = t.ok ? (int)x1[1] : 0;                          // a tap outside the image reads pixel (0, 0) and is not used
  t.fx = x1[0] - lx; t.fy = x1[1] - ly;
  t.at = (size_t)ly * cols + lx;
  return t;
}
typedef Tap4 TexelBytes;
template <class Img>
PVLM_HD inline TexelBytes tap_bytes(Img gray, int cols, const TexelTap& t) { return tap4(gray, cols, t.at); }
PVLM_HD inline float tap_value(const TexelTap& t, const TexelBytes& b) {
  const float ax = 1.f - t.fx, ay = 1.f - t.fy;
  return (b.p00 * ax + b.p01 * t.fx) * ay + (b.p10 * ax + b.p11 * t.fx) * t.fy;
}
PVLM_HD inline const float* texel_ray(const float* unit, int cols, int px, int py, int half_window, int step, int k) {
  int di, dj; texel_offset(half_window, step, k, &di, &dj);
  return unit + 3 * ((size_t)(py - half_window + di) * cols + (px - half_window + dj));
}

// Views: image(b) (the neighbour's grey image as tap4 takes it), depth[], R[][9], t[][3], n, geometric (pvlm_mvs_neighbours on the GPU)
template <class Views>
struct ColumnScorer : SerialMath, SerialFactors {
  int rows, cols, half_window, step, n, px, py;
  const float* unit; const unsigned char* ref_gray; const Views* nb; ColumnPatch P;
  // the body of ScorePixel's loop for neighbour image b: false = the image does not count (`goto next_image`, `if (nrm <= 0) continue`)
  PVLM_HD bool score_view(int b, const float* nr, float d, const float* X0, const float* factors, int n_close, float* out) const {
    const size_t ws = P.w_stride; const int ts = P.t1_stride;
    float H[9];
    homography(nb->R[b], nb->t[b], nr, d, H);
    const auto gray = nb->image(b);
    // software pipeline over the window: the ray of texel k + 2, the taps of texel k + 1 and the weight of texel k + 1 are loaded
    // while texel k is interpolated — one thread has nothing else to hide its load latency with
    float uv[3];
    { const float* r = texel_ray(unit, cols, px, py, half_window, step, 0); uv[0] = r[0]; uv[1] = r[1]; uv[2] = r[2]; }
    TexelTap cur = texel_tap(uv, rows, cols, H);
    TexelBytes cb = tap_bytes(gray, cols, cur);
    float wk = P.w[0];
    if (n > 1) { const float* r = texel_ray(unit, cols, px, py, half_window, step, 1); uv[0] = r[0]; uv[1] = r[1]; uv[2] = r[2]; }
    bool ok = true;
    float sum = 0.f;
    for (int k = 0; k < n; ++k) {
      TexelTap nxt = cur; TexelBytes nbts = cb; float wn = wk;
      if (k + 1 < n) {
        nxt = texel_tap(uv, rows, cols, H);
        nbts = tap_bytes(gray, cols, nxt);
        wn = P.w[(size_t)(k + 1) * ws];
        if (k + 2 < n) { const float* r = texel_ray(unit, cols, px, py, half_window, step, k + 2); uv[0] = r[0]; uv[1] = r[1]; uv[2] = r[2]; }
      }
      ok = ok && cur.ok;                                                              // a texel outside the neighbour image drops the image (`goto next_image`)
      const float v = tap_value(cur, cb);
      P.t1[k * ts] = v;
      sum += v * wk;                                                                  // :826-827
      cur = nxt; cb = nbts; wk = wn;
    }
    if (!ok) return false;
    float sq1 = 0.f, sq01 = 0.f;                                                      // two sums, each in index order (:830-831, :834-835)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 7                                                                      // the three loads of seven texels in flight together
#endif
    for (int k = 0; k < n; ++k) {
      const float w = P.w[k * ws], t = P.t1[k * ts] - sum;
      sq1 += t * t * w;
      sq01 += (ref_texel(ref_gray, cols, px, py, half_window, step, k) - P.mean) * w * t;
    }
    const float nrm = P.sq0 * sq1;
    if (nrm <= 0.f) return false;
    float score = sq01 / sqrtf(nrm);
    score = fminf(fmaxf(score, -1.f), 1.f);
    score = smooth_score(score, factors, n_close);
    if (nb->geometric) score = geometric_adjust(score, rows, cols, X0, nb->R[b], nb->t[b], nb->depth[b]);
    *out = score;
    return true;
  }
  PVLM_HD float operator()(const float* nr, float dep, const float* factors, int n_close) const {
    const float* u0 = unit + 3 * ((size_t)py * cols + px);
    const float X0[3] = {u0[0] * dep, u0[1] * dep, u0[2] * dep};
    const float d = X0[0] * nr[0] + X0[1] * nr[1] + X0[2] * nr[2];
    if (d > 0) return -1.f;
    float best1 = 0.f, best2 = 0.f; int count = 0;
    for (int b = 0; b < nb->n; ++b) {
      float score;
      if (!score_view(b, nr, d, X0, factors, n_close, &score)) continue;
      if (count == 0 || score > best1) { best2 = best1; best1 = score; } else if (count == 1 || score > best2) best2 = score;
      ++count;
    }
    if (count == 1) return best1;
    if (count >= 2) { float avg = 0.f; avg += best1; avg += best2; return avg / 2; }
    return -1.f;
  }
};

#if defined(__HIPCC__)
// Four threads per pixel, one neighbour image each (image b in thread b mod 4 of the pixel's quad): the per-image bodies run side by
// side and their outcomes are visited in image order, as the reference's loop visits them — the same best-two average in all four
// threads.  Everything else of process_pixel is done by the four threads alike.  Device only (quad shuffles).
template <class Views>
struct QuadScorer : ColumnScorer<Views> {
  __device__ float operator()(const float* nr, float dep, const float* factors, int n_close) const {
    const float* u0 = this->unit + 3 * ((size_t)this->py * this->cols + this->px);
    const float X0[3] = {u0[0] * dep, u0[1] * dep, u0[2] * dep};
    const float d = X0[0] * nr[0] + X0[1] * nr[1] + X0[2] * nr[2];
    if (d > 0) return -1.f;
    const int lane = (int)(threadIdx.x & 63), q = lane & 3, base = lane & ~3;
    float best1 = 0.f, best2 = 0.f; int count = 0;
    for (int b0 = 0; b0 < this->nb->n; b0 += 4) {
      float mine = 0.f; int valid = 0;
      if (b0 + q < this->nb->n) valid = this->score_view(b0 + q, nr, d, X0, factors, n_close, &mine) ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vj = __shfl(valid, base + j, 64);
        const float score = __shfl(mine, base + j, 64);
        if (!vj) continue;
        if (count == 0 || score > best1) { best2 = best1; best1 = score; } else if (count == 1 || score > best2) best2 = score;
        ++count;
      }
    }
    if (count == 1) return best1;
    if (count >= 2) { float avg = 0.f; avg += best1; avg += best2; return avg / 2; }
    return -1.f;
  }
};
#endif

}  // namespace pvlm_mvs
