// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the photometric scoring pass of the reference's panoramic PatchMatch MVS:
//   /root/reference/mvs/MVS.cpp:25-35    constants (ncc_window_size, num_texels, sigma_color, sigma_spatial, PreComputeI2C)
//   /root/reference/mvs/MVS.cpp:637-680  FillPixelPatch (bilaterally weighted reference patch)
//   /root/reference/mvs/MVS.cpp:586-618  InitConfMap (score of the current depth / normal hypothesis of every pixel)
//   /root/reference/mvs/MVS.cpp:774-923  ScorePixel — photometric term (geometric_consistency = false, no close neighbours)
//   /root/reference/mvs/MVS.cpp Sample   bilinear grey sample
//   /root/reference/sensors/Equirectangular.cpp:12-19 PreComputeI2C (ImageToCam<float> of every integer pixel)
// float32 arithmetic throughout, in the order of the cv::Matx / cv::Point3f operators the reference uses
// ([recalled] OpenCV 3.4: Matx product = left-to-right sum over k starting from 0; Point3f::dot = x*x' + y*y' + z*z').
// "parity unpinned" (no reference fixtures; OpenCV absent from this image).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "equirect.hpp"

namespace oracle {

// std::exp / std::sin / std::cos / std::acos on FLOAT arguments (mvs/MVS.cpp has `using namespace std`): the reference
// gets whatever its platform's libm returns for the float overloads — correctly rounded in glibc >= 2.40 (CORE-MATH), within
// ~0.5-1 ulp of that in older glibc, i.e. platform-dependent in the last bit.  The restatement (and the product,
// csrc/pvlm_mvs_core.h) fix the one libm-independent definition: the CORRECTLY ROUNDED float result, obtained as the double
// function rounded to float (a double result within an ulp of the exact value rounds to the same float unless the exact
// value lies within 2^-29 of a float rounding boundary).  With this — and the reference's sequential sums — the GPU path
// and the oracle agree bit for bit on the scoring pass and the PatchMatch sweep (tests/test_mvs_gpu.py).
inline float FExp(float x) { return (float)std::exp((double)x); }
inline float FSin(float x) { return (float)std::sin((double)x); }
inline float FCos(float x) { return (float)std::cos((double)x); }
inline float FAcos(float x) { return (float)std::acos((double)x); }

struct MvsView {
  int rows, cols, half_window, step;
  const uint8_t* gray;   // rows x cols
};

struct PixelPatch { std::vector<float> texels0, weight; float sq0 = 0.f; bool ok = false; };

inline int MvsNumTexels(int half_window, int step) {
  const int w = 2 * half_window + 1, q = w / step + (step > 1 ? 1 : 0);
  return q * q;
}

// MVS.cpp:637-680
inline void FillPixelPatch(const MvsView& v, int px, int py, PixelPatch& patch) {
  const int hw = v.half_window, n = MvsNumTexels(hw, v.step);
  patch.texels0.assign(n, 0.f); patch.weight.assign(n, 0.f); patch.sq0 = 0.f; patch.ok = false;
  // frame.IsInside(pt, hw, hw) with an integer point
  if (!(px >= hw && py >= hw && px < v.cols - hw && py < v.rows - hw)) return;
  const float sigma_color = -1.f / (2 * 0.2 * 0.2);                       // double expression rounded to float (:33)
  const float sigma_spatial = -1.f / (2.f * hw * hw);
  const uint8_t center = v.gray[(size_t)py * v.cols + px];
  int k = 0;
  for (int row = py - hw; row <= py + hw; row += v.step)
    for (int col = px - hw; col <= px + hw; col += v.step) {
      const uint8_t tex = v.gray[(size_t)row * v.cols + col];
      float wColor = (tex - center) / 255.f;
      wColor = wColor * wColor * sigma_color;
      const float wSpatial = ((float)((col - px) * (col - px)) + (float)((row - py) * (row - py))) * sigma_spatial;
      patch.weight[k] = FExp(wColor + wSpatial);
      patch.texels0[k] = tex;
      k++;
    }
  float sum = 0.f;
  for (int i = 0; i < n; ++i) sum += patch.weight[i];                      // std::accumulate(..., 0.f)
  for (int i = 0; i < n; ++i) patch.weight[i] /= sum;
  sum = 0;
  for (int i = 0; i < n; ++i) sum += patch.weight[i] * patch.texels0[i];
  for (int i = 0; i < n; ++i) patch.texels0[i] -= sum;
  for (int i = 0; i < n; ++i) {
    const float tmp = patch.texels0[i] * patch.weight[i];
    patch.sq0 += patch.texels0[i] * tmp;
    patch.texels0[i] = tmp;
  }
  patch.ok = !(patch.sq0 <= 1e-6);                                         // float compared with a double literal
}

// Sample(img_gray, pt): bilinear
inline float MvsSample(const uint8_t* img, int cols, float x, float y) {
  const int lx = (int)x, ly = (int)y;
  const float fx = x - lx, fy = y - ly, x1 = 1.f - fx, y1 = 1.f - fy;
  const uint8_t* p = img + (size_t)ly * cols + lx;
  return (p[0] * x1 + p[1] * fx) * y1 + (p[cols] * x1 + p[cols + 1] * fx) * fy;
}

// Sample(img, pt, functor) (mvs/MVS.cpp:1445-1467): bilinear sample of a float image that ignores the corners the functor
// rejects (each rejected corner is replaced by its neighbours in a fixed preference order); +inf when all four fail.
template <typename F>
inline float MvsSampleDepth(const float* img, int cols, float x, float y, const F& ok) {
  const int lx = (int)x, ly = (int)y;
  const float fx = x - lx, fy = y - ly, x1 = 1.f - fx, y1 = 1.f - fy;
  const float x0y0 = img[(size_t)ly * cols + lx], x1y0 = img[(size_t)ly * cols + lx + 1];
  const float x0y1 = img[(size_t)(ly + 1) * cols + lx], x1y1 = img[(size_t)(ly + 1) * cols + lx + 1];
  const bool b00 = ok(x0y0), b10 = ok(x1y0), b01 = ok(x0y1), b11 = ok(x1y1);
  if (!b00 && !b10 && !b01 && !b11) return std::numeric_limits<float>::infinity();
  return float(y1 * (x1 * (b00 ? x0y0 : (b10 ? x1y0 : (b01 ? x0y1 : x1y1))) + fx * (b10 ? x1y0 : (b00 ? x0y0 : (b11 ? x1y1 : x0y1)))) +
               fy * (x1 * (b01 ? x0y1 : (b11 ? x1y1 : (b00 ? x0y0 : x1y0))) + fx * (b11 ? x1y1 : (b01 ? x0y1 : (b10 ? x1y0 : x0y0)))));
}

// geometric-consistency adjustment of one neighbour's score (ScorePixel :857-893): forward-project X0 into the neighbour,
// read the neighbour's (photometric) depth there, back-project and penalise the angle between X0 and the returned point.
inline float GeometricAdjust(float score, const Equirectangular& eq, const float* X0, const float* R, const float* t, const float* nei_depth) {
  const float geometric_weight = 0.2; float consistency = 2;
  float X1[3];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R[3 * r + c] * X0[c]; X1[r] = s + t[r]; }
  const float depth0 = (float)std::sqrt((double)X1[0] * X1[0] + (double)X1[1] * X1[1] + (double)X1[2] * X1[2]);   // cv::norm(Point3f) -> double
  float x1[2];
  eq.CamToImage(X1, x1);
  score = 1 - score;
  if (x1[0] >= 1 && x1[1] >= 1 && x1[0] < eq.cols - 1 && x1[1] < eq.rows - 1) {
    const float depth1 = MvsSampleDepth(nei_depth, eq.cols, x1[0], x1[1], [depth0](const float& d) { return std::abs(depth0 - d) / depth0 < 0.03f; });
    if (!std::isinf(depth1)) {
      float cam[3];
      eq.ImageToCam(x1, depth1, cam);
      // R_rn = R_nr^T ; t_rn = (-R_rn) * t_nr ; X0_back = R_rn * cam + t_rn
      float t_rn[3], Xb[3];
      for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += (-R[3 * c + r]) * t[c]; t_rn[r] = s; }
      for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += R[3 * c + r] * cam[c]; Xb[r] = s + t_rn[r]; }
      // VectorAngle3D(X0, X0_back, false) in float, then * 180.0 / M_PI in double, stored as float
      float cosang = X0[0] * Xb[0] + X0[1] * Xb[1] + X0[2] * Xb[2];
      const float n1 = std::sqrt(X0[0] * X0[0] + X0[1] * X0[1] + X0[2] * X0[2]), n2 = std::sqrt(Xb[0] * Xb[0] + Xb[1] * Xb[1] + Xb[2] * Xb[2]);
      cosang /= (n1 * n2);
      const float ang = cosang >= 1.f ? 0.f : (cosang <= -1.f ? (float)M_PI : FAcos(cosang));
      const float diff_angle = ang * 180.0 / M_PI;
      consistency = std::min(diff_angle, consistency);
    }
  }
  score += geometric_weight * consistency;
  score = 1 - score;
  return std::min(1.f, std::max(-1.f, score));
}

// ScorePixel, photometric term (+ the geometric-consistency term when nei_depth != nullptr).  unit = PreComputeI2C table (rows x cols x 3 float).  R_nr / t_nr: neighbour n at
// R + 9 n / t + 3 n (row-major).  Returns the aggregated score (-1 = invalid).
// plane (4) + close (n_close NeighborPixel records): the smoothness term of ScorePixel :843-857 (propagation only).
struct NeighborPixel { float point[3]; float normal[3]; float depth; };   // mvs/MVS.h:59-64
inline float ScorePixelPhotometric(const MvsView& ref, const float* unit, int px, int py, const float* normal, float depth, const PixelPatch& patch,
                                   int n_neighbors, const uint8_t* const* nei_gray, const float* R_nr, const float* t_nr,
                                   const float* const* nei_depth = nullptr, const float* plane = nullptr, const NeighborPixel* close = nullptr, int n_close = 0) {
  const Equirectangular eq(ref.rows, ref.cols);
  const float* u0 = unit + 3 * ((size_t)py * ref.cols + px);
  const float X0[3] = {u0[0] * depth, u0[1] * depth, u0[2] * depth};
  const float d = X0[0] * normal[0] + X0[1] * normal[1] + X0[2] * normal[2];
  if (d > 0) return -1;
  const int hw = ref.half_window, w = 2 * hw + 1, n = MvsNumTexels(hw, ref.step);
  std::vector<std::pair<float, int>> score_neighbor;
  std::vector<float> texels1(n);
  for (int nb = 0; nb < n_neighbors; ++nb) {
    const float* R = R_nr + 9 * nb; const float* t = t_nr + 3 * nb;
    // H = R_nr + (1.f / d) * t_nr * normal.t()
    const float inv_d = 1.f / d;
    float H[9];
    for (int i = 0; i < 3; ++i) { const float ti = inv_d * t[i]; for (int j = 0; j < 3; ++j) H[3 * i + j] = R[3 * i + j] + ti * normal[j]; }
    int k = 0; bool outside = false;
    for (int i = 0; i < w && !outside; i += ref.step)
      for (int j = 0; j < w; j += ref.step) {
        const float* uv = unit + 3 * ((size_t)(py - hw + i) * ref.cols + (px - hw + j));
        float X1[3];
        for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += H[3 * r + c] * uv[c]; X1[r] = s; }   // Matx33f * Point3f
        float x1[2];
        eq.CamToImage(X1, x1);
        if (!(x1[0] >= 1 && x1[1] >= 1 && x1[0] < ref.cols - 1 && x1[1] < ref.rows - 1)) { outside = true; break; }   // frame.IsInside(x1, 1, 1)
        texels1[k++] = MvsSample(nei_gray[nb], ref.cols, x1[0], x1[1]);
      }
    if (outside) continue;                                                                                          // goto next_image
    float sq1 = 0, sq01 = 0, sum = 0;
    for (int i = 0; i < n; ++i) sum += texels1[i] * patch.weight[i];
    for (int i = 0; i < n; ++i) texels1[i] -= sum;
    for (int i = 0; i < n; ++i) sq1 += texels1[i] * texels1[i] * patch.weight[i];
    const float nrm = patch.sq0 * sq1;
    for (int i = 0; i < n; ++i) sq01 += patch.texels0[i] * texels1[i];
    if (nrm <= 0.f) continue;
    const float ncc = sq01 / std::sqrt(nrm);
    float score = std::min(std::max(ncc, -1.f), 1.f);
    if (n_close > 0) {
      // mvs/MVS.h:82-86
      const float smoothBonus = 0.95f, smoothBonusDepth = 1.f - smoothBonus, smoothBonusNormal = (1.f - smoothBonus) * 0.96;
      const float smoothSigmaDepth = -1.f / (2.f * 0.02f * 0.02f), smoothSigmaNormal = -1.f / (2.f * 0.22f * 0.22f);
      score = 1 - score;
      for (int q = 0; q < n_close; ++q) {
        const NeighborPixel& c = close[q];
        float diff_distance = std::abs(plane[0] * c.point[0] + plane[1] * c.point[1] + plane[2] * c.point[2] + plane[3]) / depth;   // PointToPlaneDistance(plane, point, true)
        const float factorDepth = FExp(diff_distance * diff_distance * smoothSigmaDepth);
        const float cosang = normal[0] * c.normal[0] + normal[1] * c.normal[1] + normal[2] * c.normal[2];                            // VectorAngle3D(.., .., true)
        float diff_angle = cosang >= 1.f ? 0.f : (cosang <= -1.f ? (float)M_PI : FAcos(cosang));
        const float factorNormal = FExp(diff_angle * diff_angle * smoothSigmaNormal);
        score *= (1.f - smoothBonusDepth * factorDepth) * (1.f - smoothBonusNormal * factorNormal);
      }
      score = 1 - score;
      score = std::min(1.f, std::max(-1.f, score));
    }
    if (nei_depth) score = GeometricAdjust(score, eq, X0, R, t, nei_depth[nb]);
    score_neighbor.push_back({score, nb});
  }
  if (score_neighbor.empty()) return -1;
  if (score_neighbor.size() == 1) return score_neighbor[0].first;
  std::sort(score_neighbor.begin(), score_neighbor.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; });
  float avg = 0; int count = 0;
  for (const auto& p : score_neighbor) { if (count >= 2) break; avg += p.first; count++; }
  return avg / count;
}

// InitPatchMap + InitConfMap(use_geometry = nei_depth != nullptr; nei_depth[b] = neighbour b's depth_filter): conf / depth / normal are in-out (rows x cols, rows x cols, rows x cols x 3)
inline void InitConfMap(const MvsView& ref, int n_neighbors, const uint8_t* const* nei_gray, const float* R_nr, const float* t_nr,
                        float* depth, float* normal, float* conf, const float* const* nei_depth = nullptr) {
  std::vector<float> unit((size_t)ref.rows * ref.cols * 3);
  const Equirectangular eq(ref.rows, ref.cols);
  for (int i = 0; i < ref.rows; ++i)
    for (int j = 0; j < ref.cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * ref.cols + j)]); }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int row = 0; row < ref.rows; ++row) {
    PixelPatch patch;
    for (int col = 0; col < ref.cols; ++col) {
      const size_t e = (size_t)row * ref.cols + col;
      if (depth[e] <= 0) continue;
      float c = -1;
      FillPixelPatch(ref, col, row, patch);
      if (patch.sq0 > 0) c = ScorePixelPhotometric(   /* :602 — the 1e-6 verdict of FillPixelPatch is ignored by InitPatchMap :619-629 */ ref, unit.data(), col, row, normal + 3 * e, depth[e], patch, n_neighbors, nei_gray, R_nr, t_nr, nei_depth);
      conf[e] = c;
      if (c <= -1) { depth[e] = 0; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0; }
    }
  }
}

// ProjectDepthConfToRef, depth only (mvs/MVS.cpp:2011-2070): every neighbour pixel (zero-depth ones included, as upstream)
// is moved to the reference camera and splat onto the four integer pixels around its projection; a pixel keeps the
// smallest range it receives.  out: rows x cols, 0 = nothing projected.
inline void ProjectDepthToRef(int rows, int cols, const float* unit, const float* nei_depth, const float* R_nr, const float* t_nr, float* out) {
  const Equirectangular eq(rows, cols);
  float R_rn[9], t_rn[3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_rn[3 * r + c] = R_nr[3 * c + r];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += (-R_rn[3 * r + c]) * t_nr[c]; t_rn[r] = s; }
  std::fill(out, out + (size_t)rows * cols, 0.f);
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const size_t e = (size_t)row * cols + col;
      const float pn[3] = {unit[3 * e] * nei_depth[e], unit[3 * e + 1] * nei_depth[e], unit[3 * e + 2] * nei_depth[e]};
      float pr[3];
      for (int r = 0; r < 3; ++r) { float sacc = 0; for (int c = 0; c < 3; ++c) sacc += R_rn[3 * r + c] * pn[c]; pr[r] = sacc + t_rn[r]; }
      const float range = (float)std::sqrt((double)pr[0] * pr[0] + (double)pr[1] * pr[1] + (double)pr[2] * pr[2]);
      float px[2];
      eq.CamToImage(pr, px);
      const int xs[2] = {(int)std::ceil(px[0]), (int)std::floor(px[0])}, ys[2] = {(int)std::ceil(px[1]), (int)std::floor(px[1])};
      for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
          const int x = xs[a], y = ys[b];
          if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
          float& d = out[(size_t)y * cols + x];
          if (d != 0 && d < range) continue;
          d = range;
        }
    }
}

// FilterDepthImage (mvs/MVS.cpp:1735-1790): a depth survives when at least two neighbours agree with it within 0.8 x thr
// at the pixel itself and at least five (neighbour, 4-neighbourhood) samples agree within 1.2 x thr (or the pixel is marked
// depth_constant).  depth_filter / conf_filter: rows x cols outputs (0 elsewhere); conf / depth_constant may be null.
inline void FilterDepthImage(int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* R_nr, const float* t_nr,
                             const float* depth, const float* conf, const unsigned char* depth_constant, float depth_diff_threshold,
                             float* depth_filter, float* conf_filter) {
  std::vector<float> unit((size_t)rows * cols * 3);
  const Equirectangular eq(rows, cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * cols + j)]); }
  std::vector<std::vector<float>> proj(n_neighbors, std::vector<float>((size_t)rows * cols));
  for (int b = 0; b < n_neighbors; ++b) ProjectDepthToRef(rows, cols, unit.data(), nei_depth[b], R_nr + 9 * b, t_nr + 3 * b, proj[b].data());
  const float loose = depth_diff_threshold * 1.2f, strict = depth_diff_threshold * 0.8f;
  const int ox[4] = {-1, 1, 0, 0}, oy[4] = {0, 0, 1, -1};
  std::fill(depth_filter, depth_filter + (size_t)rows * cols, 0.f);
  if (conf_filter) std::fill(conf_filter, conf_filter + (size_t)rows * cols, 0.f);
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const size_t e = (size_t)row * cols + col;
      const float d = depth[e];
      if (d <= 0) continue;
      int similar = 0;
      for (int b = 0; b < n_neighbors; ++b) { const float dn = proj[b][e]; if (dn > 0 && std::abs((d - dn) / d) < strict) similar++; }
      if (similar < 2) continue;
      similar = 0;
      for (int b = 0; b < n_neighbors; ++b)
        for (int k = 0; k < 4; ++k) {
          const int x = col + ox[k], y = row + oy[k];
          if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
          const float dn = proj[b][(size_t)y * cols + x];
          if (dn > 0 && std::abs((d - dn) / d) < loose) similar++;
        }
      const bool keep_constant = depth_constant && depth_constant[e];
      if (similar < 5 && !keep_constant) continue;
      depth_filter[e] = d;
      if (conf && conf_filter) conf_filter[e] = conf[e];
    }
}

// ProjectDepthConfToRef with project_depth = project_conf = true (mvs/MVS.cpp:2011-2070): as ProjectDepthToRef, and every
// write of a range also writes the source pixel's confidence, so a target pixel ends with the confidence of the last
// (raster order) source pixel that passed the range test.
inline void ProjectDepthConfToRef(int rows, int cols, const float* unit, const float* nei_depth, const float* nei_conf, const float* R_nr, const float* t_nr,
                                  float* out_depth, float* out_conf) {
  const Equirectangular eq(rows, cols);
  float R_rn[9], t_rn[3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_rn[3 * r + c] = R_nr[3 * c + r];
  for (int r = 0; r < 3; ++r) { float s = 0; for (int c = 0; c < 3; ++c) s += (-R_rn[3 * r + c]) * t_nr[c]; t_rn[r] = s; }
  std::fill(out_depth, out_depth + (size_t)rows * cols, 0.f);
  std::fill(out_conf, out_conf + (size_t)rows * cols, 0.f);
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const size_t e = (size_t)row * cols + col;
      const float pn[3] = {unit[3 * e] * nei_depth[e], unit[3 * e + 1] * nei_depth[e], unit[3 * e + 2] * nei_depth[e]};
      float pr[3];
      for (int r = 0; r < 3; ++r) { float sacc = 0; for (int c = 0; c < 3; ++c) sacc += R_rn[3 * r + c] * pn[c]; pr[r] = sacc + t_rn[r]; }
      const float range = (float)std::sqrt((double)pr[0] * pr[0] + (double)pr[1] * pr[1] + (double)pr[2] * pr[2]);
      float px[2];
      eq.CamToImage(pr, px);
      const int xs[2] = {(int)std::ceil(px[0]), (int)std::floor(px[0])}, ys[2] = {(int)std::ceil(px[1]), (int)std::floor(px[1])};
      for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
          const int x = xs[a], y = ys[b];
          if (!(x >= 0 && y >= 0 && x < cols && y < rows)) continue;
          float& d = out_depth[(size_t)y * cols + x];
          if (d != 0 && d < range) continue;
          d = range;
          out_conf[(size_t)y * cols + x] = nei_conf[e];
        }
    }
}

// FilterDepthImageRefine (mvs/MVS.cpp:1794-1890) — the filter the pipeline runs (mvs/MVS.cpp:194).  Confidence-weighted
// average of the agreeing depths; disagreeing neighbours subtract confidence (occlusion: the neighbour's projected
// confidence; free-space violation: the neighbour's own confidence where the reference point lands in it).  conf is
// IN-OUT (upstream zeroes frame.conf_map where depth <= 0); nei_conf[b] = neighbour b's conf_map (already through
// ConvertNCC2Conf :2343).  min_depth / max_depth: config.min_depth / config.max_depth (base/Config.h:65-66).
inline void FilterDepthImageRefine(int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* const* nei_conf, const float* R_nr,
                                   const float* t_nr, const float* depth, float* conf, const unsigned char* depth_constant, float depth_diff_threshold,
                                   float min_depth, float max_depth, float* depth_filter, float* conf_filter) {
  std::vector<float> unit((size_t)rows * cols * 3);
  const Equirectangular eq(rows, cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * cols + j)]); }
  std::vector<std::vector<float>> pd(n_neighbors, std::vector<float>((size_t)rows * cols)), pc(n_neighbors, std::vector<float>((size_t)rows * cols));
  for (int b = 0; b < n_neighbors; ++b) ProjectDepthConfToRef(rows, cols, unit.data(), nei_depth[b], nei_conf[b], R_nr + 9 * b, t_nr + 3 * b, pd[b].data(), pc[b].data());
  const float loose = depth_diff_threshold * 1.2f;
  std::fill(depth_filter, depth_filter + (size_t)rows * cols, 0.f);
  std::fill(conf_filter, conf_filter + (size_t)rows * cols, 0.f);
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const size_t e = (size_t)row * cols + col;
      const float d = depth[e];
      if (d <= 0) { conf[e] = 0; continue; }
      float positive = conf[e], negative = 0, avg = d * positive;
      int n_pos = 0, n_neg = 0;
      bool bad = false;
      for (int n = n_neighbors - 1; n >= 0; --n) {
        const float dn = pd[n][e], cn = pc[n][e];
        if (dn <= 0 && n_pos + n_neg + n < 2) { bad = true; break; }
        if (std::abs((d - dn) / d) < loose) {
          avg += dn * cn;
          positive += cn;
          n_pos += 1;
        } else {
          if (dn < d) negative += cn;
          else {
            const float X0[3] = {unit[3 * e] * d, unit[3 * e + 1] * d, unit[3 * e + 2] * d};
            const float* R = R_nr + 9 * n; const float* t = t_nr + 3 * n;
            float X1[3], x1[2];
            for (int r = 0; r < 3; ++r) { float sacc = 0; for (int c = 0; c < 3; ++c) sacc += R[3 * r + c] * X0[c]; X1[r] = sacc + t[r]; }
            eq.CamToImage(X1, x1);
            const int xr = (int)std::round(x1[0]), yr = (int)std::round(x1[1]);
            if (xr >= 0 && yr >= 0 && xr < cols && yr < rows) {
              const float c = nei_conf[n][(size_t)yr * cols + xr];
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
          continue;
        }
      }
      if (depth_constant && depth_constant[e]) { depth_filter[e] = d; conf_filter[e] = 1.f; }
    }
}


// ---- PatchMatch sweep: EstimateDepthMapSingle with Propagate::CHECKER_BOARD (mvs/MVS.cpp:682-720), PropagateCheckerBoard
// (:1098-1129), ProcessPixel (:721-772), PerturbDepthNormal3 (:1254-1320), InterpolatePixel (:1923-1935), CorrectNormal
// (:1953-1971), PerturbNormal / PerturbDepth / GenerateRandomNormal (:1368-1431).
// RANDOM DRAWS: upstream every thread of the `omp parallel for` pulls from ONE cv::RNG seeded with time(NULL)
// (mvs/MVS.cpp:30) — a data race, no two runs agree.  Here draw k of pixel e in pass p is a hash of (seed, p, e, k); the
// conversion to float is cv::RNG's (next() * 2^-32 in float; uniform(a, b) = that * (b - a) + a) [recalled, OpenCV 3.4].
inline uint32_t MvsRandomU32(uint64_t seed, uint64_t pixel, uint32_t k) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (pixel + 1) + 0xD1B54A32D192ED03ull * (uint64_t)(k + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
inline uint64_t MvsPassSeed(uint64_t seed, int pass) { return seed * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull * (uint64_t)(pass + 1); }
struct MvsRng {
  uint64_t seed, pixel; uint32_t k = 0;
  float next01() { return (float)MvsRandomU32(seed, pixel, k++) * 2.3283064365386962890625e-10f; }
  float uniform(float a, float b) { return next01() * (b - a) + a; }
};

inline float MvsDot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }   // cv::Vec3f::dot / Point3f::dot

// :1923-1935
inline float InterpolatePixel(const float* unit, int cols, int px, int py, int nx, int ny, float depth, const float* normal, float min_depth, float max_depth) {
  const float* view_ray = unit + 3 * ((size_t)py * cols + px);
  const float* un = unit + 3 * ((size_t)ny * cols + nx);
  const float X1[3] = {un[0] * depth, un[1] * depth, un[2] * depth};
  const float dnorm = MvsDot3(view_ray, normal);
  if (std::abs(dnorm) < 1e-6) return depth;
  const float depth_new = MvsDot3(X1, normal) / dnorm;
  if (depth_new >= min_depth && depth_new <= max_depth) return depth_new;
  return depth;
}

// :1953-1971.  Eigen::AngleAxisf(rad, axis).toRotationMatrix() with the UN-normalised axis upstream passes [recalled, Eigen 3.4 AngleAxis.h]
inline void CorrectNormal(const float* viewDir, float* normal) {
  const float cosAngLen = MvsDot3(normal, viewDir);
  if (cosAngLen >= 0) {
    const float axis[3] = {normal[1] * viewDir[2] - normal[2] * viewDir[1], normal[2] * viewDir[0] - normal[0] * viewDir[2], normal[0] * viewDir[1] - normal[1] * viewDir[0]};
    const float rad = std::min((FAcos(cosAngLen) - float(M_PI_2)) * 1.01f, -0.001f);
    const float sn = FSin(rad), c = FCos(rad);
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

// :1368-1396
inline void PerturbNormal(MvsRng& rng, const float* normal, float perturbation, float* out) {
  const float a1 = (rng.next01() - 0.5f) * perturbation;
  const float a2 = (rng.next01() - 0.5f) * perturbation;
  const float a3 = (rng.next01() - 0.5f) * perturbation;
  const float sin_a1 = FSin(a1), sin_a2 = FSin(a2), sin_a3 = FSin(a3);
  const float cos_a1 = FCos(a1), cos_a2 = FCos(a2), cos_a3 = FCos(a3);
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
// :1398-1403
inline float PerturbDepth(MvsRng& rng, float depth, float perturbation) {
  const float max_depth = (1 + perturbation) * depth, min_depth = (1 - perturbation) * depth;
  return rng.uniform(0.f, 1.f) * (max_depth - min_depth) + min_depth;
}
// :1405-1431
inline void GenerateRandomNormal(MvsRng& rng, const float* view_ray, float* normal) {
  float v1 = 0.0f, v2 = 0.0f, s = 2.0f;
  while (s >= 1.0f) {
    v1 = 2.0f * rng.uniform(0.f, 1.f) - 1.0f;
    v2 = 2.0f * rng.uniform(0.f, 1.f) - 1.0f;
    s = v1 * v1 + v2 * v2;
  }
  const float s_norm = std::sqrt(1.0f - s);
  normal[0] = 2.0f * v1 * s_norm; normal[1] = 2.0f * v2 * s_norm; normal[2] = 1.0f - 2.0f * s;
  if (MvsDot3(normal, view_ray) > 0) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
}

struct MvsSweep {
  MvsView ref; const float* unit; int n_neighbors; const uint8_t* const* nei_gray; const float* R_nr; const float* t_nr;
  const float* const* nei_depth;          // use_geometry when != nullptr
  const unsigned char* depth_constant;    // may be nullptr
  float min_depth, max_depth;
  float *depth, *normal, *conf;
};

// PerturbDepthNormal3 :1254-1320 (perturb_normal = true, its default)
inline bool PerturbDepthNormal3(const MvsSweep& S, MvsRng& rng, int px, int py, const PixelPatch& patch, const NeighborPixel* close, int n_close, bool perturb_depth) {
  const size_t e = (size_t)py * S.ref.cols + px;
  float* normal_origin = S.normal + 3 * e; float& depth_origin = S.depth[e]; float& conf_origin = S.conf[e];
  const float* view_ray = S.unit + 3 * e;
  const float scaleRanges[12] = {1.f, 0.5f, 0.25f, 0.125f, 0.0625f, 0.03125f, 0.015625f, 0.0078125f, 0.00390625f, 0.001953125f, 0.0009765625f, 0.00048828125f};
  float thConfSmall(0.55 * 0.2f), thConfBig(0.55 * 0.4f), thConfRand(0.55 * 0.9f);
  unsigned idxScaleRange(0);
  if (1 - conf_origin <= thConfSmall) idxScaleRange = 2;
  else if (1 - conf_origin <= thConfBig) idxScaleRange = 1;
  else if (1 - conf_origin >= thConfRand) {
    bool refine = false;
    for (int iter = 0; iter < 6; iter++) {
      float depth_random = perturb_depth ? rng.uniform(S.min_depth, S.max_depth) : depth_origin;
      float normal_random[3];
      GenerateRandomNormal(rng, view_ray, normal_random);
      const float nconf = ScorePixelPhotometric(S.ref, S.unit, px, py, normal_random, depth_random, patch, S.n_neighbors, S.nei_gray, S.R_nr, S.t_nr, S.nei_depth);
      if (nconf > conf_origin) {
        conf_origin = nconf; depth_origin = depth_random;
        normal_origin[0] = normal_random[0]; normal_origin[1] = normal_random[1]; normal_origin[2] = normal_random[2];
        if (1 - nconf < thConfRand) { refine = true; break; }
      }
    }
    if (!refine) return false;
  }
  float scaleRange(scaleRanges[idxScaleRange]);
  float depthRange = depth_origin * 0.02;
  float angleRange = 30.f / 180.f * M_PI;
  for (int iter = 0; iter < 6; iter++) {
    float depth_perturb = perturb_depth ? PerturbDepth(rng, depth_origin, scaleRange * depthRange) : depth_origin;
    float normal_perturb[3];
    PerturbNormal(rng, normal_origin, scaleRange * angleRange, normal_perturb);
    if (MvsDot3(normal_perturb, view_ray) >= 0) continue;
    const float X0[3] = {view_ray[0] * depth_perturb, view_ray[1] * depth_perturb, view_ray[2] * depth_perturb};
    const float plane[4] = {normal_perturb[0], normal_perturb[1], normal_perturb[2], -MvsDot3(normal_perturb, X0)};
    const float nconf = ScorePixelPhotometric(S.ref, S.unit, px, py, normal_perturb, depth_perturb, patch, S.n_neighbors, S.nei_gray, S.R_nr, S.t_nr, S.nei_depth, plane, close, n_close);
    if (nconf > conf_origin) {
      conf_origin = nconf; depth_origin = depth_perturb;
      normal_origin[0] = normal_perturb[0]; normal_origin[1] = normal_perturb[1]; normal_origin[2] = normal_perturb[2];
      idxScaleRange++;
      scaleRange = scaleRanges[idxScaleRange];
    }
  }
  return true;
}

// ProcessPixel :721-772; `neighbor`: the pixels whose hypotheses are propagated — the four direct neighbours PropagateCheckerBoard
// passes (:1113-1114), or the two PropagateSequential passes (:1073, :1091)
inline void ProcessPixel(const MvsSweep& S, MvsRng& rng, int px, int py, const PixelPatch& patch, int n_neighbor, const int* nx, const int* ny) {
  const int rows = S.ref.rows, cols = S.ref.cols;
  const size_t e = (size_t)py * cols + px;
  const bool keep_depth_constant = S.depth_constant && S.depth_constant[e];
  float& depth = S.depth[e]; float* normal = S.normal + 3 * e; float& conf = S.conf[e];
  NeighborPixel close[4]; int n_close = 0;
  const int cx[4] = {px - 1, px, px, px + 1}, cy[4] = {py, py - 1, py + 1, py};
  for (int q = 0; q < 4; ++q) {
    if (!(cx[q] >= 0 && cy[q] >= 0 && cx[q] < cols && cy[q] < rows)) continue;
    const size_t ne = (size_t)cy[q] * cols + cx[q];
    const float d = S.depth[ne];
    if (d <= 0) continue;
    NeighborPixel& c = close[n_close++];
    for (int k = 0; k < 3; ++k) { c.point[k] = S.unit[3 * ne + k] * d; c.normal[k] = S.normal[3 * ne + k]; }
    c.depth = d;
  }
  const float* view_ray = S.unit + 3 * e;
  for (int q = 0; q < n_neighbor; ++q) {
    if (!(nx[q] >= 0 && ny[q] >= 0 && nx[q] < cols && ny[q] < rows)) continue;
    const size_t ne = (size_t)ny[q] * cols + nx[q];
    float depth_neighbor = S.depth[ne];
    if (depth_neighbor <= 0) continue;
    float normal_neighbor[3] = {S.normal[3 * ne], S.normal[3 * ne + 1], S.normal[3 * ne + 2]};
    depth_neighbor = keep_depth_constant ? depth : InterpolatePixel(S.unit, cols, px, py, nx[q], ny[q], depth_neighbor, normal_neighbor, S.min_depth, S.max_depth);
    CorrectNormal(view_ray, normal_neighbor);
    const float X0[3] = {view_ray[0] * depth_neighbor, view_ray[1] * depth_neighbor, view_ray[2] * depth_neighbor};
    const float plane[4] = {normal_neighbor[0], normal_neighbor[1], normal_neighbor[2], -MvsDot3(normal_neighbor, X0)};
    const float newconf = ScorePixelPhotometric(S.ref, S.unit, px, py, normal_neighbor, depth_neighbor, patch, S.n_neighbors, S.nei_gray, S.R_nr, S.t_nr, S.nei_depth, plane, close, n_close);
    if (conf < newconf) { conf = newconf; depth = depth_neighbor; normal[0] = normal_neighbor[0]; normal[1] = normal_neighbor[1]; normal[2] = normal_neighbor[2]; }
  }
  PerturbDepthNormal3(S, rng, px, py, patch, close, n_close, !keep_depth_constant);
}

// EstimateDepthMapSingle(ref, CHECKER_BOARD, max_iter, conf_threshold, use_geometry = nei_depth != nullptr) :682-720.
// depth / normal / conf: in-out, already initialised (InitDepthNormal + InitConfMap).
inline void EstimateDepthMapCheckerBoard(const MvsView& ref, int n_neighbors, const uint8_t* const* nei_gray, const float* R_nr, const float* t_nr,
                                         float* depth, float* normal, float* conf, const float* const* nei_depth, const unsigned char* depth_constant,
                                         float min_depth, float max_depth, uint64_t seed, int max_iter, float conf_threshold) {
  std::vector<float> unit((size_t)ref.rows * ref.cols * 3);
  const Equirectangular eq(ref.rows, ref.cols);
  for (int i = 0; i < ref.rows; ++i)
    for (int j = 0; j < ref.cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * ref.cols + j)]); }
  const MvsSweep S{ref, unit.data(), n_neighbors, nei_gray, R_nr, t_nr, nei_depth, depth_constant, min_depth, max_depth, depth, normal, conf};
  for (int iter = 0; iter < max_iter; ++iter)
    for (int offset = 0; offset <= 1; ++offset) {
      const uint64_t pass_seed = MvsPassSeed(seed, 2 * iter + offset);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 2)
#endif
      for (int row = 0; row < ref.rows; ++row) {
        PixelPatch patch;
        for (int col = (row % 2 + offset) % 2; col < ref.cols; col += 2) {
          const size_t e = (size_t)row * ref.cols + col;
          if (depth[e] <= 0) continue;
          FillPixelPatch(ref, col, row, patch);        // frame.patch_map[e] of InitPatchMap
          if (patch.sq0 <= 1e-6) continue;
          MvsRng rng{pass_seed, (uint64_t)e};
          const int nx[4] = {col - 1, col, col + 1, col}, ny[4] = {row, row - 1, row, row + 1};
          ProcessPixel(S, rng, col, row, patch, 4, nx, ny);
        }
      }
    }
  for (int i = 0; i < ref.cols; ++i)
    for (int j = 0; j < ref.rows; ++j) {
      const size_t e = (size_t)j * ref.cols + i;
      if (depth_constant && depth_constant[e]) continue;
      if (conf[e] < conf_threshold) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0; }
    }
}

// EstimateDepthMapSingle(ref, SEQUENTIAL, max_iter, conf_threshold, use_geometry) :682-720 with PropagateSequential :1057-1097 — the
// strategy config/Room.txt:90 / config/Floor.txt:88 select (propagate_strategy = 2).  Even iterations walk the image from the top
// left in raster order and hand ProcessPixel the left and the upper neighbour; odd iterations walk back from the bottom right
// with the right and the lower neighbour.  Strictly sequential, as upstream (there the parallelism is one image per thread).
// Random draws of pixel e in iteration i: MvsRng{MvsPassSeed(seed, i), e} — the counter stream the checkerboard uses per colour pass.
inline void EstimateDepthMapSequential(const MvsView& ref, int n_neighbors, const uint8_t* const* nei_gray, const float* R_nr, const float* t_nr,
                                       float* depth, float* normal, float* conf, const float* const* nei_depth, const unsigned char* depth_constant,
                                       float min_depth, float max_depth, uint64_t seed, int max_iter, float conf_threshold) {
  std::vector<float> unit((size_t)ref.rows * ref.cols * 3);
  const Equirectangular eq(ref.rows, ref.cols);
  for (int i = 0; i < ref.rows; ++i)
    for (int j = 0; j < ref.cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * ref.cols + j)]); }
  const MvsSweep S{ref, unit.data(), n_neighbors, nei_gray, R_nr, t_nr, nei_depth, depth_constant, min_depth, max_depth, depth, normal, conf};
  PixelPatch patch;
  auto visit = [&](int col, int row, int dir, uint64_t pass_seed) {
    const size_t e = (size_t)row * ref.cols + col;
    if (depth[e] <= 0) return;                          // :1066, :1084
    FillPixelPatch(ref, col, row, patch);               // frame.patch_map[e] of InitPatchMap
    if (patch.sq0 <= 0) return;                         // :1069, :1087
    MvsRng rng{pass_seed, (uint64_t)e};
    const int nx[2] = {col + dir, col}, ny[2] = {row, row + dir};
    ProcessPixel(S, rng, col, row, patch, 2, nx, ny);
  };
  for (int iter = 0; iter < max_iter; ++iter) {
    const uint64_t pass_seed = MvsPassSeed(seed, iter);
    if (iter % 2 == 0) {
      for (int row = 0; row < ref.rows; ++row) for (int col = 0; col < ref.cols; ++col) visit(col, row, -1, pass_seed);
    } else {
      for (int row = ref.rows - 1; row >= 0; --row) for (int col = ref.cols - 1; col >= 0; --col) visit(col, row, +1, pass_seed);
    }
  }
  for (int i = 0; i < ref.cols; ++i)
    for (int j = 0; j < ref.rows; ++j) {
      const size_t e = (size_t)j * ref.cols + i;
      if (depth_constant && depth_constant[e]) continue;
      if (conf[e] < conf_threshold) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0; }
    }
}

// MVS::DepthImageToCloud (mvs/MVS.cpp:2073-2107), the per-frame body of MergeDepthImages (:2144-2166; FuseDepthMaps :224-227 saves
// MergeDepthImages(2) as MVS-merge.pcd): every pixel with 0 < depth < 0.8 max_depth becomes the world point
// TranslatePoint<float, double>(ImageToCam(col, row) * depth, T_wc) (base/Geometry.hpp:545-551) with the pixel's colour, unless the
// colour is "sky blue" (BGR2HSV, util/Visualization.cpp:57-77: H in [100, 124], S in [43, 200], V in [150, 255]).
// Raster order.  xyz: n x 3 float, rgb: n x 3 uint8 (r, g, b).  Returns n.
inline void Bgr2Hsv(const unsigned char* bgr, float* hsv) {
  const float r = bgr[2] / 255.f, g = bgr[1] / 255.f, b = bgr[0] / 255.f;
  const float C_max = std::max(r, std::max(g, b)), C_min = std::min(r, std::min(g, b));
  if (C_max == 0) { hsv[0] = hsv[1] = hsv[2] = 0; return; }
  const float delta_C = C_max - C_min;
  float h = 0.f;
  if (C_max == r) h = 60.f * ((g - b) / delta_C + 6 * (g < b));
  else if (C_max == g) h = 60.f * ((b - r) / delta_C + 2);
  else if (C_max == b) h = 60.f * ((r - g) / delta_C + 4);
  hsv[0] = h / 360.f; hsv[1] = delta_C / C_max; hsv[2] = C_max;
}
// normal / normal_out != null and filter_sky = false: MVS::DepthNormalToCloud (mvs/MVS.cpp:2109-2142), normal_world = R_wc * Vector3d(n).
inline long long DepthImageToCloud(int rows, int cols, const float* depth, const unsigned char* bgr, const double* T_wc, float max_depth, float* xyz,
                                   unsigned char* rgb, bool filter_sky = true, const float* normal = nullptr, float* normal_out = nullptr) {
  const Equirectangular eq(rows, cols);
  long long n = 0;
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const float d = depth[(size_t)row * cols + col];
      if (d <= 0 || d >= max_depth * 0.8) continue;
      const float px[2] = {(float)col, (float)row};
      float ray[3];
      eq.ImageToCam(px, 1.f, ray);
      const float pc[3] = {ray[0] * d, ray[1] * d, ray[2] * d};
      float pw[3];
      for (int k = 0; k < 3; ++k) pw[k] = (float)(pc[0] * T_wc[4 * k] + pc[1] * T_wc[4 * k + 1] + pc[2] * T_wc[4 * k + 2] + T_wc[4 * k + 3]);
      const unsigned char* c = bgr + 3 * ((size_t)row * cols + col);
      float hsv[3];
      Bgr2Hsv(c, hsv);
      hsv[0] *= 180.f; hsv[1] *= 255.f; hsv[2] *= 255.f;
      if (filter_sky && hsv[0] >= 100.f && hsv[0] <= 124.f && hsv[1] >= 43.f && hsv[1] <= 200.f && hsv[2] >= 150.f && hsv[2] <= 255.f) continue;
      if (normal_out) {
        const float* nc = normal + 3 * ((size_t)row * cols + col);
        for (int k = 0; k < 3; ++k) normal_out[3 * n + k] = (float)(T_wc[4 * k] * (double)nc[0] + T_wc[4 * k + 1] * (double)nc[1] + T_wc[4 * k + 2] * (double)nc[2]);
      }
      xyz[3 * n] = pw[0]; xyz[3 * n + 1] = pw[1]; xyz[3 * n + 2] = pw[2];
      rgb[3 * n] = c[2]; rgb[3 * n + 1] = c[1]; rgb[3 * n + 2] = c[0];
      ++n;
    }
  return n;
}

// MVS::FuseDepthImages(use_filtered_depth = true) (mvs/MVS.cpp:2168-2334; the second half of FuseDepthMaps :229-230, MVS-fuse.pcd): the
// confidence-weighted fusion "from openMVS".  Frames are visited by decreasing neighbour count (std::sort, :2189); a pixel with a valid
// depth that nobody has claimed claims the pixels it projects to in its neighbours when their depths agree (relative difference <
// depth_diff_threshold), averages position and colour with weights ConfToWeight (:2337-2340), and becomes a cloud point when >= 2 neighbour
// pixels agreed; otherwise its claims are withdrawn.  Neighbour depths in front of which the point lies (|X1| < n_depth) are zeroed in
// the neighbour's map once the point is accepted.  A literal restatement, upstream's state machine included:
//   * view_project / invalid_depth live OUTSIDE the frame loop and are cleared at the end of a pixel's body — which the early `continue`s
//     (invalid depth, claimed pixel, and the sky-colour rejection :2316) skip, so their contents carry over to the next processed pixel;
//   * every frame starts with neighbours + 1 references on its depth map, one is dropped per visit as reference or neighbour (:2322-2331),
//     at zero the map is released; a released map is read again from the depth file when the frame is needed as somebody's NEIGHBOUR or
//     is itself the reference's id in the read loop (:2209-2215; the file holds the UNFILTERED estimate, and the zeroing is lost), but a
//     reference whose map is gone is skipped (:2205-2206, the test precedes the read loop);
//   * the colour test runs on the rounded colour (cv::Vec3b(color): saturate_cast), the stored colour is truncated (static_cast<uchar>).
// depth_filter[i] (may be null = empty) is the working map and is MODIFIED; depth_saved[i] (may be null) is what ReadFrameDepth would load;
// present_after / maps_after (n, n x rows x cols; optional): which frames hold a depth_filter map after the call, and its contents.
// conf[i] = the <id>_filter.bin confidence; bgr[i] the colour image; T_wc[i] 16 doubles.  nei_off (n + 1) / nei / R_nr (9 each) / t_nr (3 each).
struct FusedPoint { float x, y, z; unsigned char r, g, b; };
inline float ConfToWeight(float conf, float depth) { return 1.f / (std::max(1.f - conf, 0.03f) * (depth * depth)); }
inline std::vector<FusedPoint> FuseDepthImages(int n, int rows, int cols, float* const* depth_filter, const float* const* depth_saved, const float* const* conf,
                                               const unsigned char* const* bgr, const double* T_wc, const int* frame_id, const int* nei_off, const int* nei,
                                               const float* R_nr, const float* t_nr, float max_depth, float depth_diff_threshold,
                                               int* present_after = nullptr, float* maps_after = nullptr) {
  const Equirectangular eq(rows, cols);
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(3 * npix);                                                      // PreComputeI2C
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) { const float px[2] = {(float)j, (float)i}; eq.ImageToCam(px, 1.f, &unit[3 * ((size_t)i * cols + j)]); }
  std::vector<FusedPoint> cloud;
  std::vector<std::vector<uint16_t>> occupied(n, std::vector<uint16_t>(npix, 65535));
  std::vector<float*> map(n);                                                             // frames[i].depth_filter.data, null = empty
  std::vector<std::vector<float>> reloaded(n);
  for (int i = 0; i < n; ++i) map[i] = depth_filter[i];
  std::vector<std::pair<int, int>> idx_connections;
  std::vector<int> depth_count(n);
  for (int i = 0; i < n; ++i) { idx_connections.push_back({i, nei_off[i + 1] - nei_off[i]}); depth_count[i] = nei_off[i + 1] - nei_off[i] + 1; }
  std::sort(idx_connections.begin(), idx_connections.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second > b.second; });
  std::vector<std::pair<size_t, std::pair<int, int>>> view_project, invalid_depth;       // (frame, (x, y))
  auto translate = [](const float* p, const double* T, float* o) {                      // TranslatePoint<float, double>
    for (int k = 0; k < 3; ++k) o[k] = (float)(p[0] * T[4 * k] + p[1] * T[4 * k + 1] + p[2] * T[4 * k + 2] + T[4 * k + 3]);
  };
  for (const std::pair<int, int>& pc : idx_connections) {
    const int ref_idx = pc.first;
    std::vector<int> ids = {ref_idx};
    for (int b = nei_off[ref_idx]; b < nei_off[ref_idx + 1]; ++b) ids.push_back(nei[b]);
    if (map[ref_idx] != nullptr) {
      for (int id : ids)                                                                  // :2209-2215
        if (map[id] == nullptr && depth_saved[id] != nullptr) { reloaded[id].assign(depth_saved[id], depth_saved[id] + npix); map[id] = reloaded[id].data(); }
      const float* ref_depth_map = map[ref_idx];
      for (int row = 0; row < rows; row++)
        for (int col = 0; col < cols; col++) {
          const size_t e = (size_t)row * cols + col;
          const float depth = ref_depth_map[e];
          if (depth <= 0 || depth >= max_depth * 0.8) continue;
          uint16_t& occupied_id = occupied[ref_idx][e];
          if (occupied_id != 65535) continue;
          occupied_id = (uint16_t)frame_id[ref_idx];
          float color[3] = {(float)bgr[ref_idx][3 * e], (float)bgr[ref_idx][3 * e + 1], (float)bgr[ref_idx][3 * e + 2]};
          float confidence = ConfToWeight(conf[ref_idx][e], depth);
          const float X0[3] = {unit[3 * e] * depth, unit[3 * e + 1] * depth, unit[3 * e + 2] * depth};
          float X[3];
          translate(X0, T_wc + 16 * (size_t)ref_idx, X);
          for (int k = 0; k < 3; ++k) { X[k] = X[k] * confidence; color[k] = color[k] * confidence; }
          for (int b = nei_off[ref_idx]; b < nei_off[ref_idx + 1]; ++b) {
            const size_t n_idx = (size_t)nei[b];
            if (map[n_idx] == nullptr) continue;
            const float* R = R_nr + 9 * (size_t)b; const float* t = t_nr + 3 * (size_t)b;
            float X1[3];
            for (int r = 0; r < 3; ++r) { float sacc = 0; for (int c = 0; c < 3; ++c) sacc += R[3 * r + c] * X0[c]; X1[r] = sacc + t[r]; }   // cv::Matx33f * Point3f + Point3f
            float x1[2];
            eq.CamToImage(X1, x1);
            const int px = (int)std::round(x1[0]), py = (int)std::round(x1[1]);
            if (!(px >= 0 && py >= 0 && px < cols && py < rows)) continue;
            const size_t ne = (size_t)py * cols + px;
            const float n_depth = map[n_idx][ne];
            if (n_depth <= 0) continue;
            uint16_t& n_occupied_id = occupied[n_idx][ne];
            if (n_occupied_id != 65535) continue;
            if (std::abs((depth - n_depth) / depth) < depth_diff_threshold) {
              view_project.push_back({n_idx, {px, py}});
              const float n_confidence = ConfToWeight(conf[n_idx][ne], n_depth);
              const float n_point[3] = {unit[3 * ne] * n_depth, unit[3 * ne + 1] * n_depth, unit[3 * ne + 2] * n_depth};
              float Xn[3];
              translate(n_point, T_wc + 16 * n_idx, Xn);
              for (int k = 0; k < 3; ++k) { X[k] += Xn[k] * n_confidence; color[k] += (float)bgr[n_idx][3 * ne + k] * n_confidence; }
              confidence += n_confidence;
              n_occupied_id = occupied_id;
            }
            if (std::sqrt((double)X1[0] * X1[0] + (double)X1[1] * X1[1] + (double)X1[2] * X1[2]) < n_depth) invalid_depth.push_back({n_idx, {px, py}});   // cv::norm(Point3f)
          }
          if (view_project.size() < 2) {
            for (const auto& p : view_project) occupied[p.first][(size_t)p.second.second * cols + p.second.first] = 65535;
            occupied_id = 65535;
          } else {
            const float nrm = 1.f / confidence;
            const float point_world[3] = {X[0] * nrm, X[1] * nrm, X[2] * nrm};
            for (int k = 0; k < 3; ++k) color[k] = color[k] * nrm;
            FusedPoint p;
            p.x = point_world[0]; p.y = point_world[1]; p.z = point_world[2];
            p.r = (unsigned char)color[2]; p.g = (unsigned char)color[1]; p.b = (unsigned char)color[0];
            for (const auto& q : invalid_depth)
              if (map[q.first] != nullptr) map[q.first][(size_t)q.second.second * cols + q.second.first] = 0;
            unsigned char rounded[3];                                                     // cv::Vec3b(color): saturate_cast<uchar>(float) = clamp(cvRound)
            for (int k = 0; k < 3; ++k) { const long v = std::lrint(color[k]); rounded[k] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
            float hsv[3];
            Bgr2Hsv(rounded, hsv);
            hsv[0] *= 180.f; hsv[1] *= 255.f; hsv[2] *= 255.f;
            if (hsv[0] >= 100 && hsv[0] <= 124 && hsv[1] >= 43 && hsv[1] <= 200 && hsv[2] >= 150 && hsv[2] <= 255) continue;   // NB: skips the clears below
            cloud.push_back(p);
          }
          invalid_depth.clear();
          view_project.clear();
        }
    }
    for (int id : ids)                                                                    // next_image: :2322-2331
      if (--depth_count[id] <= 0) map[id] = nullptr;                                      // depth_filter.release()
  }
  for (int i = 0; i < n && present_after; ++i) {                                          // the frames' depth_filter members after the call
    present_after[i] = map[i] != nullptr;
    if (map[i] && maps_after) std::copy(map[i], map[i] + npix, maps_after + (size_t)i * npix);
  }
  return cloud;
}

// MVS::SelectNeighborKNN (mvs/MVS.cpp:334-382).  valid[i], R_wc (9, row-major), t_wc (3) per frame.  Output: for every frame the
// list of (neighbour id, R_nr float 9, t_nr float 3).  KdTreeFLANN::nearestKSearch restated as a brute-force sorted float32
// search (ties by index); T_nr = T_wn^-1 * T_wr through the general 4x4 inverse like upstream (Gauss-Jordan with partial
// pivoting here; Eigen uses a cofactor expansion [recalled] — both exact to ~1e-16 for a rigid transform).
struct MvsNeighbor { int id; float R_nr[9]; float t_nr[3]; };
inline void Inverse4(const double* A, double* out) {
  double M[4][8];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { M[r][c] = A[4 * r + c]; M[r][4 + c] = r == c ? 1.0 : 0.0; }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r) if (std::abs(M[r][col]) > std::abs(M[piv][col])) piv = r;
    for (int c = 0; c < 8; ++c) std::swap(M[col][c], M[piv][c]);
    const double inv = 1.0 / M[col][col];
    for (int c = 0; c < 8; ++c) M[col][c] *= inv;
    for (int r = 0; r < 4; ++r) if (r != col) { const double f = M[r][col]; for (int c = 0; c < 8; ++c) M[r][c] -= f * M[col][c]; }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = M[r][4 + c];
}
inline std::vector<std::vector<MvsNeighbor>> SelectNeighborKNN(int n, const int* valid, const double* R_wc, const double* t_wc, int neighbor_size, float sq_distance_threshold) {
  std::vector<std::vector<MvsNeighbor>> neighbors(n);
  std::vector<int> owner;
  for (int i = 0; i < n; ++i) if (valid[i]) owner.push_back(i);
  auto pose = [&](int i, double* T) {
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4 * r + c] = R_wc[9 * i + 3 * r + c]; T[4 * r + 3] = t_wc[3 * i + r]; }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
  };
  for (int ref = 0; ref < n; ++ref) {
    if (!valid[ref]) continue;
    const float q[3] = {(float)t_wc[3 * ref], (float)t_wc[3 * ref + 1], (float)t_wc[3 * ref + 2]};
    std::vector<std::pair<float, int>> cand;
    for (int j : owner) {
      const float c[3] = {(float)t_wc[3 * j], (float)t_wc[3 * j + 1], (float)t_wc[3 * j + 2]};
      float sq = 0; for (int m = 0; m < 3; ++m) { const float diff = q[m] - c[m]; sq += diff * diff; }
      cand.push_back({sq, j});
    }
    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    if ((int)cand.size() > neighbor_size * 3) cand.resize(neighbor_size * 3);
    for (size_t i = 1; i < cand.size() && (int)neighbors[ref].size() < neighbor_size; ++i) {
      if (cand[i].first < sq_distance_threshold) continue;
      double Tn[16], Tr[16], Tni[16], T[16];
      pose(cand[i].second, Tn); pose(ref, Tr);
      Inverse4(Tn, Tni);
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double acc = 0; for (int m = 0; m < 4; ++m) acc += Tni[4 * r + m] * Tr[4 * m + c]; T[4 * r + c] = acc; }
      MvsNeighbor nb; nb.id = cand[i].second;
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) nb.R_nr[3 * r + c] = (float)T[4 * r + c]; nb.t_nr[r] = (float)T[4 * r + 3]; }
      neighbors[ref].push_back(nb);
    }
  }
  return neighbors;
}

// MVS::InitDepthNormal (mvs/MVS.cpp:496-584), the branch the reference compiles (`#elif 1`, :511-514): the LiDAR depth image
// of ProjectLidar2PanoramaDepth(cloud, rows, cols, T_cl, 2) (uint16, depth * 256) seeds the depth map, every other pixel
// gets a uniform random depth (rng.fill(UNIFORM, max_depth, min_depth), :546), keep_lidar_constant marks the seeded pixels,
// the mask zeroes excluded pixels, and every pixel the mask keeps gets GenerateRandomNormal.  lidar16 == nullptr is the
// `use_lidar == false` branch (:566-569).  Random draws: counter based like the sweep (MvsRng; upstream: one time-seeded
// cv::RNG) — draw 0 of pixel e is its random depth, the following draws its normal.
inline void InitDepthNormal(int rows, int cols, const unsigned short* lidar16, const float* mask, float min_depth, float max_depth, bool keep_lidar_constant,
                            uint64_t seed, float* depth, float* normal, unsigned char* depth_constant) {
  const Equirectangular eq(rows, cols);
  const uint64_t ps = MvsPassSeed(seed, -2);
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      const size_t e = (size_t)row * cols + col;
      MvsRng rng{ps, (uint64_t)e};
      float d = lidar16 ? (float)lidar16[e] : 0.f;                       // convertTo(CV_32F)
      d /= 256.f;
      const float depth_random = rng.next01() * (min_depth - max_depth) + max_depth;   // uniform(a = max_depth, b = min_depth)
      const float lidar_mask = d > 0 ? 0.f : 1.f;                        // cv::threshold(depth, 0, 1, THRESH_BINARY_INV)
      d = d + depth_random * lidar_mask;                                 // cv::add(depth, random.mul(lidar_mask))
      if (lidar16 && keep_lidar_constant && depth_constant) depth_constant[e] = (unsigned char)(1.f - lidar_mask);
      const float m = mask ? mask[e] : 1.f;
      depth[e] = d * m;                                                  // depth_map.mul(mask)
      normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0.f;
      if (m < 1) continue;
      const float p[2] = {(float)col, (float)row};
      float ray[3];
      eq.ImageToCam(p, 1.f, ray);
      GenerateRandomNormal(rng, ray, normal + 3 * e);
    }
}

// MVS::RemoveSmallSegments (mvs/MVS.cpp:1504-1577): region growing over the 4-neighbourhood in COLUMN-major seed order; a
// neighbour joins the region of the pixel it is reached from when its depth is positive and within depth_diff_threshold
// (relative to the depth of the pixel it is reached FROM — the relation is not symmetric, so the seed order matters);
// regions smaller than min_segment lose depth, normal and confidence.  Pixels without depth are regions of size one.
inline int RemoveSmallSegments(int rows, int cols, float depth_diff_threshold, int min_segment, float* depth, float* normal, float* conf) {
  std::vector<unsigned char> done((size_t)rows * cols, 0);
  std::vector<int> seg_list((size_t)rows * cols);
  int removed = 0;
  for (int u = 0; u < cols; u++)
    for (int v = 0; v < rows; v++) {
      if (done[(size_t)v * cols + u]) continue;
      seg_list[0] = v * cols + u;
      unsigned seg_list_count = 1, seg_list_curr = 0;
      while (seg_list_curr < seg_list_count) {
        const int cur = seg_list[seg_list_curr];
        const int cx = cur % cols, cy = cur / cols;
        const int nx[4] = {cx - 1, cx + 1, cx, cx}, ny[4] = {cy, cy, cy - 1, cy + 1};
        const float depth_curr = depth[cur];
        for (int i = 0; i < 4; i++) {
          if (!(nx[i] >= 0 && ny[i] >= 0 && nx[i] < cols && ny[i] < rows)) continue;       // frame.IsInside
          const int nb = ny[i] * cols + nx[i];
          if (!done[nb]) {
            const float depth_neighbor = depth[nb];
            if (depth_neighbor > 0 && std::abs((depth_curr - depth_neighbor) / depth_curr) < depth_diff_threshold) {
              seg_list[seg_list_count++] = nb;
              done[nb] = 1;
            }
          }
        }
        ++seg_list_curr;
        done[cur] = 1;
      }
      if (seg_list_count < (unsigned)min_segment)
        for (unsigned i = 0; i < seg_list_count; i++) {
          const int e = seg_list[i];
          depth[e] = 0; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0; conf[e] = -1;
          removed++;
        }
    }
  return removed;
}

}  // namespace oracle
