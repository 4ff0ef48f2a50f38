// pvlm_host_internal.hpp — what the translation units of the host mirror share besides pvlm_host.hpp: the stage timers of the tools' tables,
// the small dense helpers, and the few helpers one reference file's mirror needs from another's.  Not installed; not part of the interface.
#pragma once
#include "pvlm_host.hpp"
#include "../csrc/pvlm_workers.h"
#include "../csrc/pvlm_undistort_core.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iomanip>
#include <limits>
#include <map>
#include <mutex>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace pvlm {

// seconds and calls per named stage of the mirrored calls (tools/*_like_*.py print them); defined in pvlm_host.cpp
std::map<std::string, double>& Stages();
std::map<std::string, long>& StageCallCounts();
struct StageTimer {
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~StageTimer() { Stages()[name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++StageCallCounts()[name]; }
};

// small dense helpers (row-major 3x3)
static inline Vector3d MatVec(const Matrix3d& R, const Vector3d& p) {
  return {(R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2], (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]};
}
static inline Vector3d MatTVec(const Matrix3d& R, const Vector3d& p) {
  return {(R[0] * p[0] + R[3] * p[1]) + R[6] * p[2], (R[1] * p[0] + R[4] * p[1]) + R[7] * p[2], (R[2] * p[0] + R[5] * p[1]) + R[8] * p[2]};
}


// plane angles of the association filters
static inline double PlaneAngle(const double* a, const double* b) {  // base/Geometry.hpp:471-485
  double c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  c = c / (std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
  return c >= 1.0 ? 0.0 : std::acos(c);
}

static inline double PlaneAngleN(const double* a, const double* b) {  // PlaneAngle(..., normalized = true)
  const double c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  return c >= 1.0 ? 0.0 : std::acos(c);
}

// the equirectangular camera model of the panorama frames (sfm/PanoramaFrame + camera model of the reference)
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
struct Equirect {
  int cols, rows;
  template <typename T> void ImageToCam(const T* px, T r, T* cam) const {
    T sx = (2 * px[0] / cols - 1) * M_PI;
    T sy = (0.5 - px[1] / rows) * M_PI;
    T cy = (T)std::cos((double)sy);
    cam[0] = r * cy * (T)std::sin((double)sx);
    cam[1] = -r * (T)std::sin((double)sy);
    cam[2] = r * cy * (T)std::cos((double)sx);
  }
  template <typename T> void CamToImage(const T* cam, T* px) const {
    T lon = FastAtan2(cam[0], cam[2]);
    T lat = -FastAtan2(cam[1], (T)std::sqrt(cam[0] * cam[0] + cam[2] * cam[2]));
    px[0] = cols * (0.5 + lon / (2.0 * M_PI));
    px[1] = rows * (0.5 - lat / M_PI);
  }
  std::vector<float> BreakToSegments(const float* start, const float* end, float seg_length) const {
    float p1[3], p2[3];
    ImageToCam(start, 5.0f, p1);
    ImageToCam(end, 5.0f, p2);
    const float sl[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const float length = std::sqrt((start[0] - end[0]) * (start[0] - end[0]) + (start[1] - end[1]) * (start[1] - end[1]));
    const int count = length / seg_length + 1;
    std::vector<float> seg = {start[0], start[1]};
    for (int i = 1; i < count; i++) {
      const float f = i * 1.f / count;
      const float p[3] = {p1[0] + f * sl[0], p1[1] + f * sl[1], p1[2] + f * sl[2]};
      float pixel[2];
      CamToImage(p, pixel);
      const float lastx = seg[seg.size() - 2];
      if (std::abs(pixel[0] - lastx) > 0.8 * cols) {
        const float gq = p1[0] / (p1[0] - p2[0]);
        const float q[3] = {p1[0] + gq * sl[0], p1[1] + gq * sl[1], p1[2] + gq * sl[2]};
        float left[2];
        CamToImage(q, left);
        left[0] = 0;
        const float right[2] = {float(cols - 1), left[1]};
        if (pixel[0] > lastx) { seg.insert(seg.end(), {left[0], left[1], right[0], right[1]}); }
        else { seg.insert(seg.end(), {right[0], right[1], left[0], left[1]}); }
      }
      seg.push_back(pixel[0]); seg.push_back(pixel[1]);
    }
    seg.push_back(end[0]); seg.push_back(end[1]);
    return seg;
  }
};

}  // namespace pvlm
