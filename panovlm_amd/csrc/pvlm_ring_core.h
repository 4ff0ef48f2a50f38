// Per-element bodies of the range-image kernels (csrc/pvlm_ring.hip, K16-K22): the device side of Velodyne::ReOrderVLP
// (sensors/Velodyne.cpp:371-526), Velodyne::Segmentation (:1438-1586) and the adaptive-window curvature (:623-657).
// Host/device so that tests/cpp/ring_core_check.cpp can drive the very same functions serially on a machine without a GPU
// (tests/test_ring_core_cpu.py: every array equals the CPU restatement of the reference).  libpvlm.so has no host path.
//
// Arithmetic contract: every float expression is the reference's, operation for operation; the including translation unit is
// compiled with -ffp-contract=off.  Decisions that pass through a FLOAT libm call of the reference's host (atan, atan2) are
// taken for the whole interval of floats within kUlps of the true value (computed in fp64): see the header of pvlm_ring.hip.
#pragma once
#include <cmath>
#include <cstring>

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__ __forceinline__
#endif

namespace pvlm_ring {

constexpr int kUlps = 4;
constexpr int kMaxRings = 64;

struct RingScan {
  long long pt0;     // first point of the scan in the batch's point arrays
  long long cell0;   // first cell of the scan in the batch's cell arrays
  int n;             // raw points
  int pad;
  double start_ori;  // azimuth of the first raw point in [0, 2 pi) — std::atan2f of the host (:397-399)
};
struct Point { float x, y, z, w; };

// k steps along the ordered line of floats (+0 and -0 share a place)
PVLM_HD float step_ulps(float f, int k) {
  unsigned u; memcpy(&u, &f, 4);
  const int i = (int)u;
  int o = i >= 0 ? i : (int)(0x80000000u - (unsigned)i);
  o += k;
  const unsigned r = o >= 0 ? (unsigned)o : 0x80000000u - (unsigned)o;
  float g; memcpy(&g, &r, 4);
  return g;
}

// sensors/Velodyne.cpp:170-211, `deg` already rounded to float (:439)
PVLM_HD int ring_of_elevation(float deg, int rings) {
  if (!(deg == deg)) return -1;
  int id = -1;
  if (rings == 16) {
    id = (int)((double)((deg + 15.0f) / 2.0f) + 0.5);
    if (id > 15 || id < 0) id = -1;
  } else if (rings == 32) {
    id = (int)(((double)deg + 92.0 / 3.0) * 3.0 / 4.0);
    if (id > 31 || id < 0) id = -1;
  } else if (rings == 64) {
    id = (double)deg >= -8.83 ? (int)((double)(2.0f - deg) * 3.0 + 0.5) : 32 + (int)((-8.83 - (double)deg) * 2.0 + 0.5);
    if ((double)deg > 2.0 || (double)deg < -24.33 || id > 50 || id < 0) id = -1;
  }
  return id;
}
// float atan() result -> ring: `atan(...) * 180 / M_PI` assigned to a float (:439)
PVLM_HD int ring_of_atan(float a, int rings) { return ring_of_elevation((float)((double)(a * 180.0f) / M_PI), rings); }
// float atan2() result -> azimuth in [0, 2 pi) as a double (:444-446)
PVLM_HD double ori_of_atan2(float f) { double a = (double)f; if (a < 0) a += 2 * M_PI; return a; }
PVLM_HD int column_of(double ori, bool wrapped, double start_ori, double column_width) {
  ori += 2 * M_PI * (wrapped ? 1 : 0);
  return (int)round((ori - start_ori) / column_width);
}
// position of a ring in the VLP-16 firing sequence (the std::map of :407-414; a missing key reads as 0)
PVLM_HD int firing_slot(int ring, int rings) { return (rings != 16 || ring < 0) ? 0 : (ring <= 7 ? 2 * ring : 2 * ring - 15); }
// the argument of the elevation's atan (:439), NaN for a return on the sensor's vertical axis at the origin
PVLM_HD float elevation_tangent(float x, float y, float z) { return -y / sqrtf(x * x + z * z); }

// ---- K16: ring and azimuth of one raw point with their certificates --------------------------------------------------------
// *az = the float nearest to the true atan2(x, z); *ring = the ring every float within kUlps of the true atan() gives.  Returns
// true when the point must be listed: some float of the interval gives another ring, or another column (in either state of
// the +z crossing), or the + 2 pi of :445-446 is undecided.
PVLM_HD bool classify_point(const RingScan& sc, int rings, int horizon, float x, float y, float z, float* az, int* ring) {
  bool list = false;
  const float q = elevation_tangent(x, y, z);
  int r = -1;
  if (q == q) {
    const float a = (float)atan((double)q);
    r = ring_of_atan(a, rings);
    for (int k = -kUlps; k <= kUlps; ++k) list |= ring_of_atan(step_ulps(a, k), rings) != r;
  }
  const float f = (float)atan2((double)x, (double)z);
  const float lo = step_ulps(f, -kUlps), hi = step_ulps(f, kUlps);
  if (lo < 0.f && !(hi < 0.f)) list = true;
  else if (r >= 0 || list) {
    const double w = 2.0 * M_PI / horizon;
    const double olo = ori_of_atan2(lo), ohi = ori_of_atan2(hi);
    list |= column_of(olo, false, sc.start_ori, w) != column_of(ohi, false, sc.start_ori, w);
    list |= column_of(olo, true, sc.start_ori, w) != column_of(ohi, true, sc.start_ori, w);
  }
  *az = f; *ring = r;
  return list;
}

// ---- K17: the column state machine of :431-507 for one scan ------------------------------------------------------------------
// The loop carries five scalars from point to point (crossed, last azimuth, column offset, last column, last ring): replayed as
// written over the per-point values of K16.  `exact[i]` != 0: az[i] is the host libm's own atan2f (interval of one float).
// col[i] = column or -1 (rejected), pos[i] = position of the point inside its ring; count(r) = reference to the ring's counter.
// Returns -1, or the index of the point at which the +z crossing (:447-461) could not be certified, with *last_point = the point whose
// azimuth is `last_ori` there: the caller makes the azimuths of those two points and of the N_SCANS points behind the first exact
// (host libm) and replays the scan — with exact values on both sides the comparison is the reference's own.
template <class Counter>
PVLM_HD int columns_scan(const RingScan& sc, int rings, int horizon, const float* az, const signed char* ring, const unsigned char* exact, int* col_pos /* 2 per point */,
                         Counter&& count, int* last_point) {
  const double w = 2.0 * M_PI / horizon;
  bool wrapped = false;
  double last_lo = -1, last_hi = -1;
  int shift = 0, prev_col = 0, prev_ring = -1, last_i = -1;
  *last_point = -1;
  for (int i = 0; i < sc.n; ++i) {
    const int r = ring[i];
    if (r < 0) { col_pos[2 * i] = -1; col_pos[2 * i + 1] = 0; continue; }
    const float f = az[i];
    const double lo = ori_of_atan2(exact[i] ? f : step_ulps(f, -kUlps)), hi = ori_of_atan2(exact[i] ? f : step_ulps(f, kUlps));
    if (!wrapped && lo < last_hi) {                                  // `ori < last_ori` (:447) is possible
      const bool sure = hi < last_lo;
      int sure_behind = 0, maybe_behind = 0, seen = 0;
      for (int j = i + 1; j < i + rings + 1 && j < sc.n; ++j) {
        const float g = az[j];
        const double jlo = ori_of_atan2(exact[j] ? g : step_ulps(g, -kUlps)), jhi = ori_of_atan2(exact[j] ? g : step_ulps(g, kUlps));
        sure_behind += jhi < last_lo ? 1 : 0;
        maybe_behind += jlo < last_hi ? 1 : 0;
        ++seen;
        if (maybe_behind < seen) break;                               // one return is certainly not behind: `reliable` cannot reach N_SCANS
      }
      if (sure && sure_behind >= rings) wrapped = true;
      else if (maybe_behind >= rings) { *last_point = last_i; return i; }
    }
    int col = column_of(lo, wrapped, sc.start_ori, w);               // == the column of `hi` (K16 listed the point otherwise, and it is exact now)
    if (firing_slot(r, rings) < firing_slot(prev_ring, rings)) {     // a new firing sequence started: same column as the last one?
      shift = prev_col == col ? 1 : 0;
      prev_col = col + shift;
    }
    prev_ring = r;
    col += shift;
    while (col >= horizon) col -= horizon;
    if (col < 0) { col_pos[2 * i] = -1; col_pos[2 * i + 1] = 0; continue; }
    col_pos[2 * i] = col; col_pos[2 * i + 1] = count(r)++;
    const double turn = 2 * M_PI * (wrapped ? 1 : 0);
    last_lo = lo + turn; last_hi = hi + turn; last_i = i;
  }
  return -1;
}

// ---- K19: "joined" between two 4-neighbours of the range image (:1500-1512): 1 / 0, or -1 = undecided (y, x for the host) ------
PVLM_HD int joined_certified(float a, float b, float s, float c, float theta, float* y_out, float* x_out) {
  const float far = fmaxf(a, b), near = fminf(a, b);
  if (near == 0.f) return 0;                                   // atan2(+0, x >= 0) = +0 in every libm (C Annex F): not > theta
  const float y = near * s, x = far - near * c;
  *y_out = y; *x_out = x;
  const float f = (float)atan2((double)y, (double)x);
  const bool lo = step_ulps(f, -kUlps) > theta, hi = step_ulps(f, kUlps) > theta;
  return lo == hi ? (lo ? 1 : 0) : -1;
}
// a component survives with >= 30 cells, or >= 5 cells over >= 3 rows counting pushed cells only (:1533-1543)
PVLM_HD bool keep_component(int size, int rows_with_pushed_cells) { return size >= 30 || (size >= 5 && rows_with_pushed_cells >= 3); }

// ---- K22: adaptive-window curvature of kept point i (:623-657) -------------------------------------------------------------------
// Kept as upstream, including the right-hand walk guarded by the LEFT index and the window test that looks at the left end twice;
// where upstream would read past the end of the cloud (undefined behaviour) the walk stops and the point has no curvature.
// P: the kept cloud (n points, w = ring), range: cloudDistance, [lo, hi] = [scanStartInd, scanEndInd] of the point's ring.
PVLM_HD float dist2(const Point& a, const Point& b) {   // base/Geometry.hpp:38-40
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}
PVLM_HD void curvature_point(const Point* P, const float* range, int n, int lo, int hi, int i, float* curvature, int* half_window) {
  *curvature = -1.f; *half_window = -1;
  if (hi - lo < 5 || i < lo || i > hi) return;
  const Point pi = P[i];
  int a = i - 5, e = i + 5;
  while (a >= lo && (double)dist2(P[a], pi) < 0.0064) --a;
  while (a <= hi && e < n && (double)dist2(P[e], pi) < 0.0064) ++e;
  const int h = (i - a) > (e - i) ? (i - a) : (e - i);
  a = i - h; e = i + h;
  if (a < lo - 5 || a > hi + 5 || e >= n) return;
  float acc = 0;
  for (int k = a; k <= e; ++k) acc += range[k];
  acc -= (float)(e - a + 1) * range[i];
  acc /= (float)(e - a);
  *curvature = fabsf(acc); *half_window = h;
}

}  // namespace pvlm_ring
