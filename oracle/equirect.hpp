// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the spherical camera model:
//   /root/reference/base/Math.h:15-29                 FastAtan2 (7th-order odd minimax, DOUBLE literals
//                                                     -> for T=float the polynomial is evaluated in
//                                                     double and rounded to float on assignment)
//   /root/reference/sensors/Equirectangular.h:42-72   CamToSphere
//   /root/reference/sensors/Equirectangular.h:81-96   SphereToImage
//   /root/reference/sensors/Equirectangular.h:99-114  ImageToSphere
//   /root/reference/sensors/Equirectangular.h:117-146 SphereToCam
//   /root/reference/sensors/Equirectangular.h:149-182 ImageToCam / CamToImage
//   /root/reference/sensors/Equirectangular.h:184-204 IsInside
//   /root/reference/sensors/Equirectangular.cpp:20-65 BreakToSegments
// FastAtan2 IS pinned against the real reference: oracle/_ref/libref_math.so is compiled from
// /root/reference/base/Math.h itself (oracle/Makefile) and compared bit-for-bit in
// tests/test_oracle_ref_math.py; golden vectors generated from it live in tests/golden/.
// The rest of this file is "parity unpinned" (needs OpenCV/Eigen to build the reference).
// float trig (cos/sin of a float) is evaluated as (float)f((double)x); whether the reference
// binds ::cos(float) to cosf or to cos(double) depends on header inclusion order upstream and
// differs by <= 1 ulp(float) — documented in DESIGN.md.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace oracle {

template <typename T>
inline T FastAtan2(const T& y, const T& x) {
  T ax = std::abs(x), ay = std::abs(y);
  T a = std::min(ax, ay) / (std::max(ax, ay) + (T)DBL_EPSILON);
  T s = a * a;
  T r = ((-0.04432655554792128 * s + 0.1555786518463281) * s - 0.3258083974640975) * s * a + 0.9997878412794807 * a;
  if (ay > ax) r = M_PI_2 - r;
  if (x < 0) r = M_PI - r;
  if (y < 0) r = -r;
  return r;
}

struct Equirectangular {
  int cols, rows;
  Equirectangular(int _rows, int _cols) : cols(_cols), rows(_rows) {}

  template <typename T> void CamToSphere(const T* p, T* s) const {
    s[0] = FastAtan2(p[0], p[2]);
    s[1] = -FastAtan2(p[1], (T)std::sqrt(p[0] * p[0] + p[2] * p[2]));
  }
  template <typename T> void SphereToImage(const T* s, T* px) const {
    px[0] = cols * (0.5 + s[0] / (2.0 * M_PI));
    px[1] = rows * (0.5 - s[1] / M_PI);
  }
  template <typename T> void ImageToSphere(const T* px, T* s) const {
    s[0] = (2 * px[0] / cols - 1) * M_PI;
    s[1] = (0.5 - px[1] / rows) * M_PI;
  }
  template <typename T> void SphereToCam(const T* s, T r, T* cam) const {
    T cy = (T)std::cos((double)s[1]);
    cam[0] = r * cy * (T)std::sin((double)s[0]);
    cam[1] = -r * (T)std::sin((double)s[1]);
    cam[2] = r * cy * (T)std::cos((double)s[0]);
  }
  template <typename T> void ImageToCam(const T* px, T r, T* cam) const { T s[2]; ImageToSphere(px, s); SphereToCam(s, r, cam); }
  template <typename T> void CamToImage(const T* cam, T* px) const { T s[2]; CamToSphere(cam, s); SphereToImage(s, px); }
  bool IsInside(float x, float y) const { return x >= 0 && y >= 0 && x < cols && y < rows; }
  bool IsInsideI(int x, int y) const { return x >= 0 && y >= 0 && x + 1 <= cols && y + 1 <= rows; }

  // Equirectangular.cpp:20-65 — returns flat (u,v) pairs.
  std::vector<float> BreakToSegments(const float* start, const float* end, float seg_length) const {
    float p1[3], p2[3];
    ImageToCam(start, 5.0f, p1);
    ImageToCam(end, 5.0f, p2);
    float sl[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    float length = std::sqrt((start[0] - end[0]) * (start[0] - end[0]) + (start[1] - end[1]) * (start[1] - end[1]));
    int count = length / seg_length + 1;
    std::vector<float> seg = {start[0], start[1]};
    for (int i = 1; i < count; i++) {
      const float f = i * 1.f / count;
      float p[3] = {p1[0] + f * sl[0], p1[1] + f * sl[1], p1[2] + f * sl[2]};
      float pixel[2];
      CamToImage(p, pixel);
      const float lastx = seg[seg.size() - 2];
      if (std::abs(pixel[0] - lastx) > 0.8 * cols) {
        const float g = p1[0] / (p1[0] - p2[0]);
        float q[3] = {p1[0] + g * sl[0], p1[1] + g * sl[1], p1[2] + g * sl[2]};
        float left[2];
        CamToImage(q, left);
        left[0] = 0;
        float right[2] = {float(cols - 1), left[1]};
        if (pixel[0] > lastx) { seg.push_back(left[0]); seg.push_back(left[1]); seg.push_back(right[0]); seg.push_back(right[1]); }
        else { seg.push_back(right[0]); seg.push_back(right[1]); seg.push_back(left[0]); seg.push_back(left[1]); }
      }
      seg.push_back(pixel[0]); seg.push_back(pixel[1]);
    }
    seg.push_back(end[0]); seg.push_back(end[1]);
    return seg;
  }
};

}  // namespace oracle
