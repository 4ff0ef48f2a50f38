// K24: the picks of Velodyne::ExtractFeatures on the device — ExtractEdgeFeatures2 (sensors/Velodyne.cpp:883-1000), ExtractPlaneFeatures2
// (:1098-1189) and the pcl::VoxelGrid that thins a ring's less-flat points (:1177-1181) — one wave per ring.  Included by pvlm_ring.hip.
//
// A ring is a chain: a pick disables its neighbours (non-maximum suppression, :969-986 / :1140-1155), the next candidate must still be alive,
// the plane picks read the states the edge picks left, and the less-flat list of a sector reads the states after that sector's plane picks.  The
// chain is walked as upstream walks it — sector by sector, candidates in the order K23 left them — by a wave whose lanes share each step: the
// candidates of a sector are fetched and pre-judged 64 at a time (curvature window, incidence angle), the suppression scans 64 neighbours at a
// time, the less-flat filter and the voxel sums are ordered compactions.  What is serial is what upstream's results hang on.
//
// Decisions that pass through the host's float libm: the incidence angle `acos(|a . (l - r)| / (range |l - r|)) * 180 / pi` against the
// threshold (:935-945).  As everywhere in this file (pvlm_ring_core.h) it is taken for every float within kUlps of the true acos (fp64); a
// candidate whose interval straddles the threshold makes the ring UNDECIDED when the walk reaches it alive, and the scan goes through the
// host's PickFeatures.  The same for a ring with a sector K23 left to the host, with more points than the launch provided LDS for, with a
// voxel grid beyond int32 cells (PCL passes the input through) or when the batch's centroid buffer is full.
//
// std::sort of the (voxel, point) pairs by voxel alone (PCL's cloud_point_index_idx): pvlm_stdsort.h on packed words in LDS, by the whole wave (sort_wave).
#pragma once
#include "pvlm_ring_core.h"
#include "pvlm_stdsort.h"

namespace pvlm_ring {

constexpr int kCornerSlots = 6 * 30, kFlatSlots = 6 * 4;     // per ring: <= 30 edge picks and <= 4 plane picks in each of six sectors
constexpr unsigned char kNormal = 0x01, kLessSharp = 0x02, kSharp = 0x04, kFlat = 0x08, kGround = 0x10, kDisable = 0x20;   // PointClassification

// incidence-angle test of an edge candidate: 0 = rejected, 1 = accepted, 2 = the host libm decides
PVLM_HD int view_angle_verdict(const Point& a, const Point& l, const Point& r, float range, float threshold) {
  const float bx = l.x - r.x, by = l.y - r.y, bz = l.z - r.z;
  const float along = a.x * bx + (a.y * by + a.z * bz);
  const float blen = sqrtf(bx * bx + (by * by + bz * bz));
  const float q = fabsf(along) / (range * blen);
  const double t = acos((double)q);
  if (t != t) return 1;                                       // acos of > 1 (or NaN): both comparisons of :943 are false, the candidate stays
  const float mid = (float)t;
  int seen = 0;
  for (int k = -kUlps; k <= kUlps; ++k) {
    float v = step_ulps(mid, k);
    v = (float)((double)v * (180.0 / M_PI));
    seen |= (v < threshold || v > 180 - threshold) ? 1 : 2;
  }
  return seen == 1 ? 0 : (seen == 2 ? 1 : 2);
}

struct PickArrays {
  const RingScan* scans; const int* ring_count2; const int* counts;
  const float4* cloud2; const float* range2; const float* curvature; const int* half_window; const int* order; const unsigned char* sector_host;
  unsigned char* state;          // per point
  int* corner;                   // per ring: count, then kCornerSlots entries: point | sharp << 31
  int* flat;                     // per ring: count, then kFlatSlots points
  int2* voxel_span;              // per ring: (first, count) in `voxels`
  float4* voxels; int* voxel_counter; int voxel_cap;
  unsigned char* ring_host;      // per ring: 1 = undecided / out of bounds -> the scan's picks run on the host
};

#if defined(__HIPCC__)
__device__ inline unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline int first_bit(unsigned long long m) { return m ? __ffsll((long long)m) - 1 : 64; }
__device__ inline unsigned long long below(int b) { return b >= 64 ? ~0ull : ((1ull << b) - 1ull); }

// non-maximum suppression around a picked point (:969-986, :1140-1155): lanes test 64 neighbours at a time
__device__ inline void suppress_wave(const Point* P, unsigned char* st, int begin, int lo, int hi, int ind, int lane) {
  const Point c = P[ind];
  for (int l0 = 1;; l0 += 64) {
    const int l = l0 + lane, p = ind + l;
    bool stop = p > hi;
    if (!stop) stop = l <= 5 ? (double)dist2(P[p], P[p - 1]) > 0.05 : (double)dist2(P[p], c) > 0.0036;
    const int first = first_bit(ballot64(stop));
    if (lane < first) st[p - begin] |= kDisable;
    if (first < 64) break;
  }
  for (int l0 = 1;; l0 += 64) {
    const int l = l0 + lane, p = ind - l;
    bool stop = p < lo;
    if (!stop) stop = l <= 5 ? (double)dist2(P[p], P[p + 1]) > 0.05 : (double)dist2(P[p], c) > 0.0036;
    const int first = first_bit(ballot64(stop));
    if (lane < first) st[p - begin] |= kDisable;
    if (first < 64) break;
  }
}

// grid (rings, scans), 64 threads; dynamic LDS (pick_lds_bytes): (cap + kFlatSlots) x 8 (the less-flat list: point indices, then (voxel, point) words), the
// scratch of the sort, cap x 1 (states)
inline size_t pick_lds_bytes(int cap) {
  const size_t list_cap = (size_t)cap + kFlatSlots;
  return list_cap * 8 + (2 * pvlm_stdsort::kWaveQueue + pvlm_stdsort::kWideStack + (list_cap + 31) / 32 + 1 + 4) * 4 + list_cap * 4 + (size_t)cap;
}
__global__ __launch_bounds__(64) void k_ring_picks(PickArrays A, int rings, int cap, float max_curvature, float angle_threshold) {
  extern __shared__ unsigned long long lds_words[];
  const int list_cap = cap + kFlatSlots;
  unsigned long long* key = lds_words;           // less-flat list: the point index in the low half; the voxel joins it in the high half once the bounds are known
  unsigned* queue = reinterpret_cast<unsigned*>(key + list_cap);                 // scratch of pvlm_stdsort::sort_by_levels
  unsigned* cuts = queue + 2 * pvlm_stdsort::kWaveQueue + pvlm_stdsort::kWideStack;
  int* ctr = reinterpret_cast<int*>(cuts + (list_cap + 31) / 32 + 1);
  unsigned short* pos = reinterpret_cast<unsigned short*>(ctr + 4);
  unsigned char* st = reinterpret_cast<unsigned char*>(pos + 2 * list_cap);
  const int s = blockIdx.y, ring = blockIdx.x, lane = threadIdx.x;
  const size_t slot = (size_t)s * rings + ring;
  const RingScan sc = A.scans[s];
  const int n = A.counts[2 * s + 1];
  int begin = 0;
  for (int r = 0; r < ring; ++r) begin += A.ring_count2[(size_t)s * kMaxRings + r];
  const int cnt = A.ring_count2[(size_t)s * kMaxRings + ring];
  const int lo = begin + 5, hi = begin + cnt - 6, span = hi - lo;
  int* corner = A.corner + slot * (1 + kCornerSlots);
  int* flat = A.flat + slot * (1 + kFlatSlots);
  if (lane == 0) { corner[0] = 0; flat[0] = 0; A.voxel_span[slot] = make_int2(0, 0); A.ring_host[slot] = 0; }
  if (n == 0 || cnt == 0) return;
  const Point* P = reinterpret_cast<const Point*>(A.cloud2 + sc.pt0);
  const float* curv = A.curvature + sc.pt0;
  const int* order = A.order + sc.pt0;
  unsigned char* state_out = A.state + sc.pt0 + begin;
  if (span < 6) { for (int i = lane; i < cnt; i += 64) state_out[i] = kNormal; return; }        // the picks skip such a ring (:707-723)
  bool refuse = cnt > cap;
  for (int j = 0; j < 6; ++j) refuse |= A.sector_host[slot * 6 + j] != 0;
  if (refuse) { if (lane == 0) A.ring_host[slot] = 1; return; }
  for (int i = lane; i < cnt; i += 64) st[i] = kNormal;
  __syncthreads();

  // ---- ExtractEdgeFeatures2: per sector from the largest curvature down, <= 3 sharp + 27 less sharp -----------------------------------
  int n_corner = 0;
  bool undecided = false;
  for (int j = 0; j < 6 && !undecided; ++j) {
    const int sp = lo + span * j / 6, ep = lo + span * (j + 1) / 6 - 1;
    int picked = 0;
    bool done = false;
    for (int base = 0; !done && base <= ep - sp; base += 64) {
      const int k = ep - base - lane;
      const bool valid = k >= sp;
      const int ind = valid ? order[k] : begin;
      const float c = valid ? curv[ind] : -1.f;
      const bool dead = !valid || (double)c < 0.1;              // sorted: everything further down the sector is below 0.1 too
      int verdict = 0;
      if (!dead && !(c > max_curvature)) {
        const int h = A.half_window[sc.pt0 + ind];
        verdict = view_angle_verdict(P[ind], P[ind - h], P[ind + h], A.range2[sc.pt0 + ind], angle_threshold);
      }
      const int first_dead = first_bit(ballot64(dead));
      unsigned long long todo = ballot64(verdict != 0) & below(first_dead);
      while (todo) {
        const int i = first_bit(todo);
        todo &= todo - 1;
        const int ind_i = __shfl(ind, i), v = __shfl(verdict, i);
        if (st[ind_i - begin] != kNormal) continue;
        if (v == 2) { undecided = true; done = true; break; }
        ++picked;
        if (picked > 30) { done = true; break; }
        if (lane == 0) {
          corner[1 + n_corner] = ind_i | (picked <= 3 ? (int)0x80000000u : 0);
          st[ind_i - begin] = picked <= 3 ? kSharp : kLessSharp;
        }
        ++n_corner;
        __syncthreads();
        suppress_wave(P, st, begin, lo, hi, ind_i, lane);
        __syncthreads();
      }
      if (first_dead < 64) done = true;
    }
  }
  if (undecided) { if (lane == 0) A.ring_host[slot] = 1; return; }
  if (lane == 0) corner[0] = n_corner;

  // ---- ExtractPlaneFeatures2: per sector from the smallest curvature up, <= 4 flat; then the sector's less-flat points ---------------
  int n_flat = 0, n_list = 0;
  for (int j = 0; j < 6; ++j) {
    const int sp = lo + span * j / 6, ep = lo + span * (j + 1) / 6 - 1;
    int picked = 0;
    bool done = false;
    for (int base = 0; !done && picked < 4 && base <= ep - sp; base += 64) {
      const int k = sp + base + lane;
      const bool valid = k <= ep;
      const int ind = valid ? order[k] : begin;
      const bool dead = !valid || (double)curv[ind] > 0.02;     // sorted: everything further up the sector is above 0.02 too
      const int first_dead = first_bit(ballot64(dead));
      unsigned long long todo = below(first_dead);
      while (todo && picked < 4) {
        const int i = first_bit(todo);
        todo &= todo - 1;
        const int ind_i = __shfl(ind, i);
        const unsigned char was = st[ind_i - begin];
        if (was != kNormal && was != kGround) continue;
        if (lane == 0) {
          flat[1 + n_flat] = ind_i;
          if (was == kNormal) key[n_list] = (unsigned)ind_i;
          st[ind_i - begin] = was | kFlat;
        }
        ++n_flat; ++picked;
        if (was == kNormal) ++n_list;
        __syncthreads();
        suppress_wave(P, st, begin, lo, hi, ind_i, lane);
        __syncthreads();
      }
      if (first_dead < 64) done = true;
    }
    // :1157-1163 — k is a point index here, not a position in the sorted order (as upstream)
    for (int k0 = sp; k0 <= ep; k0 += 64) {
      const int k = k0 + lane;
      bool take = false;
      if (k <= ep) { const unsigned char v = st[k - begin]; take = (v & kNormal) && !(v & kDisable) && (double)curv[k] < 0.3; }
      const unsigned long long m = ballot64(take);
      if (take) key[n_list + __popcll(m & below(lane))] = (unsigned)k;
      n_list += __popcll(m);
    }
    __syncthreads();
  }
  if (lane == 0) flat[0] = n_flat;
  for (int i = lane; i < cnt; i += 64) state_out[i] = st[i];
  if (n_list == 0) return;

  // ---- pcl::VoxelGrid, leaf 0.2 (restated as host/pvlm_features.cpp VoxelGridAppend restates it) ---------------------------------------
  const float inv = 1.f / 0.2f;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = lane; q < n_list; q += 64) {
    const Point p = P[(unsigned)key[q]];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  for (int d = 32; d > 0; d >>= 1)
    for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], d)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d)); }
  long long cells = 1;
  for (int a = 0; a < 3; ++a) cells *= (long long)((mx[a] - mn[a]) * inv) + 1;
  if (cells > 2147483647ll) { if (lane == 0) A.ring_host[slot] = 1; return; }
  int vbase[3], vdim[3];
  for (int a = 0; a < 3; ++a) { vbase[a] = (int)floorf(mn[a] * inv); vdim[a] = (int)floorf(mx[a] * inv) - vbase[a] + 1; }
  const int stride_y = vdim[0], stride_z = vdim[0] * vdim[1];
  for (int q = lane; q < n_list; q += 64) {
    const unsigned pt = (unsigned)key[q];
    const Point p = P[pt];
    const int cx = (int)(floorf(p.x * inv) - (float)vbase[0]);
    const int cy = (int)(floorf(p.y * inv) - (float)vbase[1]);
    const int cz = (int)(floorf(p.z * inv) - (float)vbase[2]);
    key[q] = ((unsigned long long)(unsigned)(cx + cy * stride_y + cz * stride_z) << 32) | pt;
  }
  __syncthreads();
  const bool sane = pvlm_stdsort::sort_wave(key, n_list, [](unsigned long long x, unsigned long long y) { return (unsigned)(x >> 32) < (unsigned)(y >> 32); }, queue, cuts,
                                            ctr, pos, lane);
  if (!sane) { if (lane == 0) A.ring_host[slot] = 1; return; }
  // voxels = runs of equal cells; a run's sum is taken in the sorted order, by one lane
  int n_vox = 0;
  for (int q0 = 0; q0 < n_list; q0 += 64) {
    const int q = q0 + lane;
    const bool start = q < n_list && (q == 0 || (unsigned)(key[q] >> 32) != (unsigned)(key[q - 1] >> 32));
    n_vox += __popcll(ballot64(start));
  }
  __shared__ int first_out;
  if (lane == 0) first_out = atomicAdd(A.voxel_counter, n_vox);
  __syncthreads();
  const int out0 = first_out;
  if (out0 + n_vox > A.voxel_cap) { if (lane == 0) A.ring_host[slot] = 1; return; }
  int seen = 0;
  for (int q0 = 0; q0 < n_list; q0 += 64) {
    const int q = q0 + lane;
    const bool start = q < n_list && (q == 0 || (unsigned)(key[q] >> 32) != (unsigned)(key[q - 1] >> 32));
    const unsigned long long m = ballot64(start);
    if (start) {
      const unsigned cell = (unsigned)(key[q] >> 32);
      float sx = 0, sy = 0, sz = 0;
      int e = q;
      for (; e < n_list && (unsigned)(key[e] >> 32) == cell; ++e) { const Point p = P[(unsigned)key[e]]; sx += p.x; sy += p.y; sz += p.z; }
      const float w = (float)(e - q);
      A.voxels[out0 + seen + __popcll(m & below(lane))] = make_float4(sx / w, sy / w, sz / w, (float)kNormal);
    }
    seen += __popcll(m);
  }
  if (lane == 0) A.voxel_span[slot] = make_int2(out0, n_vox);
}
#endif

}  // namespace pvlm_ring
