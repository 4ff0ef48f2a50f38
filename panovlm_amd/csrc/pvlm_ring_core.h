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
constexpr int kRingGroup = 8;                          // rings whose in-ring positions columns_block counts in one walk
constexpr int kColumnsScratch = 6 + 2 * kRingGroup;   // ints of scratch per thread of columns_block

struct RingScan {
  long long pt0;     // first point of the scan in the batch's point arrays
  long long cell0;   // first cell of the scan in the batch's cell arrays
  int n;             // raw points
  int pad;
  long long slot0;   // first slot of the scan in the chunk-transposed arrays (rec, colpos): chunk_slots(n) slots
  double start_ori;  // azimuth of the first raw point in [0, 2 pi) — std::atan2f of the host (:397-399)
};
struct Point { float x, y, z, w; };
// K16's result for one raw point: azimuth (float atan2) and ring (-1 = none) with the flag "az / ring are the host libm's own"
struct PointRec { float az; int ring_exact; };
PVLM_HD PointRec make_rec(float az, int ring, bool exact) { PointRec q; q.az = az; q.ring_exact = (ring & 0xFF) | (exact ? 0x100 : 0); return q; }
PVLM_HD int rec_ring(const PointRec& q) { return (int)(signed char)(q.ring_exact & 0xFF); }
PVLM_HD bool rec_exact(const PointRec& q) { return (q.ring_exact & 0x100) != 0; }
// storage slot of point i of a scan cut into `threads` chunks of `chunk` consecutive points (columns_block): chunk-transposed
constexpr int kColumnThreads = 1024;
PVLM_HD int chunk_of(int n, int threads) { return (n + threads - 1) / threads; }
PVLM_HD int chunk_slot(int i, int chunk, int threads) { return (i % chunk) * threads + i / chunk; }
PVLM_HD long long chunk_slots(int n, int threads) { return (long long)chunk_of(n, threads) * threads; }   // slots a scan of n points occupies

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
// written over the per-point values of K16.  rec_exact: the azimuth is the host libm's own atan2f (interval of one float).
// col[i] = column or -1 (rejected), pos[i] = position of the point inside its ring; count(r) = reference to the ring's counter.
// Returns -1, or the index of the point at which the +z crossing (:447-461) could not be certified, with *last_point = the point whose
// azimuth is `last_ori` there: the caller makes the azimuths of those two points and of the N_SCANS points behind the first exact
// (host libm) and replays the scan — with exact values on both sides the comparison is the reference's own.
template <class Counter>
PVLM_HD int columns_scan(const RingScan& sc, int rings, int horizon, const PointRec* rec /* natural order */, int* col_pos /* 2 per point */, Counter&& count,
                         int* last_point) {
  const double w = 2.0 * M_PI / horizon;
  bool wrapped = false;
  double last_lo = -1, last_hi = -1;
  int shift = 0, prev_col = 0, prev_ring = -1, last_i = -1;
  *last_point = -1;
  for (int i = 0; i < sc.n; ++i) {
    const int r = rec_ring(rec[i]);
    if (r < 0) { col_pos[2 * i] = -1; col_pos[2 * i + 1] = 0; continue; }
    const float f = rec[i].az;
    const bool exact_i = rec_exact(rec[i]);
    const double lo = ori_of_atan2(exact_i ? f : step_ulps(f, -kUlps)), hi = ori_of_atan2(exact_i ? f : step_ulps(f, kUlps));
    if (!wrapped && lo < last_hi) {                                  // `ori < last_ori` (:447) is possible
      const bool sure = hi < last_lo;
      int sure_behind = 0, maybe_behind = 0, seen = 0;
      for (int j = i + 1; j < i + rings + 1 && j < sc.n; ++j) {
        const float g = rec[j].az;
        const bool exact_j = rec_exact(rec[j]);
        const double jlo = ori_of_atan2(exact_j ? g : step_ulps(g, -kUlps)), jhi = ori_of_atan2(exact_j ? g : step_ulps(g, kUlps));
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

// ---- K17, one WORKGROUP per scan: the same state machine in parallel ------------------------------------------------------------
// columns_scan above is a loop of n dependent steps (measured: 1.7 us per point with one lane per scan, 48 ms for a Room batch).
// Its state decomposes:
//   * the column offset (:476-481) only changes at the points where a new firing sequence starts ("events", known from the rings
//     alone); with c_e the raw column of event e, shift_e = (c_{e-1} + shift_{e-1} == c_e), i.e. shift_e = f_e(shift_{e-1}) with
//     f_e = NOT when c_e == c_{e-1}, IDENTITY when c_e == c_{e-1} + 1, CONST 0 otherwise (c_0 + shift_0 = 0): a prefix composition
//     of functions {0,1} -> {0,1}, associative;
//   * `last_ori` is the azimuth of the last ACCEPTED point, a take-the-right-most prefix; the crossing test of point i (:447-461)
//     then reads the N_SCANS points behind i only, so every point can be tested independently, and the FIRST point whose test does
//     not come out "no crossing" decides (crossed there, or undecided there);
//   * before the crossing nothing depends on it, so one pass assumes "not crossed", finds the crossing point W, and — only when
//     there is one — a second pass recomputes columns and offsets with `crossed` = (i >= W);
//   * the position of a point inside its ring is a per-ring prefix count of the accepted points.
// Every thread owns a contiguous chunk of the scan; chunk summaries are combined by block-wide scans.  Exec abstracts the
// workgroup: threads(), and phase(f) = run f(t) for every thread t, then a barrier (the host check runs the t loop serially).
// Scratch: kColumnsScratch arrays of threads() ints (LDS on the device).  Results are those of columns_scan, bit for bit.
namespace detail {
constexpr int kFnConst0 = 0, kFnNot = 1, kFnId = 2;                  // bit s of the code = f(s)
PVLM_HD int fn_apply(int f, int s) { return (f >> s) & 1; }
PVLM_HD int fn_then(int first, int then) { return fn_apply(then, fn_apply(first, 0)) | (fn_apply(then, fn_apply(first, 1)) << 1); }
PVLM_HD int fn_of_columns(int c_prev_plus_shift_free, int c) { return c == c_prev_plus_shift_free ? kFnNot : (c == c_prev_plus_shift_free + 1 ? kFnId : kFnConst0); }
constexpr int kNone = -0x7fffffff - 1;
PVLM_HD int take_right(int l, int r) { return r != kNone ? r : l; }
PVLM_HD int take_min(int l, int r) { return l < r ? l : r; }
PVLM_HD int take_sum(int l, int r) { return l + r; }
// inclusive scan of a[0 .. T) with `op` (Hillis-Steele through tmp)
template <class Exec, class Op>
PVLM_HD void block_scan(Exec& ex, int* a, int* tmp, Op op) {
  const int T = ex.threads();
  for (int off = 1; off < T; off <<= 1) {
    ex.phase([&](int t) { tmp[t] = t >= off ? op(a[t - off], a[t]) : a[t]; });
    ex.phase([&](int t) { a[t] = tmp[t]; });
  }
}
}  // namespace detail

template <class Exec>
PVLM_HD int columns_block(Exec& ex, const RingScan& sc, int rings, int horizon, const PointRec* rec, int* col_pos, int* ring_count /* kMaxRings */, int* last_point,
                          int* scratch /* kColumnsScratch x threads() */) {
  using namespace detail;
  const int T = ex.threads(), n = sc.n;
  const int L = (n + T - 1) / T;
  // storage order of rec / col_pos: point i = t L + k of chunk t sits in slot k T + t (chunk_slot), so that step k of every thread's walk
  // reads T consecutive records — with the natural order a wave's load touched 64 cache lines and the walks were 80 % of the kernel
  auto slot = [&](int i) { return chunk_slot(i, L, T); };
  int* prev_ring_in = scratch;            // ring of the last ring-valid point before the chunk (-1: none)
  int* ev_last = scratch + T;             // raw column of the last event in / before the chunk
  int* fn = scratch + 2 * T;              // composed offset function of / up to the chunk
  int* acc_last = scratch + 3 * T;        // last accepted point in / before the chunk
  int* tmp = scratch + 4 * T;
  int* aux = scratch + 5 * T;             // first-event column, candidate point, per-ring count
  const double w = 2.0 * M_PI / horizon;
  auto lo_rec = [&](const PointRec& q) { return ori_of_atan2(rec_exact(q) ? q.az : step_ulps(q.az, -kUlps)); };
  auto hi_rec = [&](const PointRec& q) { return ori_of_atan2(rec_exact(q) ? q.az : step_ulps(q.az, kUlps)); };
  auto lo_of = [&](int i) { return lo_rec(rec[slot(i)]); };
  auto hi_of = [&](int i) { return hi_rec(rec[slot(i)]); };
  // ring of the last ring-valid point before each chunk
  ex.phase([&](int t) {
    int last = kNone;
    for (int k = 0, i = t * L; k < L && i < n; ++k, ++i) { const int r = rec_ring(rec[k * T + t]); if (r >= 0) last = r; }
    prev_ring_in[t] = last;
  });
  block_scan(ex, prev_ring_in, tmp, take_right);
  ex.phase([&](int t) { tmp[t] = t > 0 && prev_ring_in[t - 1] != kNone ? prev_ring_in[t - 1] : -1; });
  ex.phase([&](int t) { prev_ring_in[t] = tmp[t]; });
  // columns and acceptance with `crossed` = (i >= W)
  auto shift_pass = [&](int W) {
    ex.phase([&](int t) {                 // events of the chunk: first / last raw column, composition of all but the first
      int pr = prev_ring_in[t], first = kNone, last = kNone, g = kFnId;
      for (int k = 0, i = t * L; k < L && i < n; ++k, ++i) {
        const PointRec q = rec[k * T + t];
        const int r = rec_ring(q);
        if (r < 0) continue;
        if (firing_slot(r, rings) < firing_slot(pr, rings)) {
          const int c = column_of(lo_rec(q), i >= W, sc.start_ori, w);
          if (first == kNone) first = c; else g = fn_then(g, fn_of_columns(last, c));
          last = c;
        }
        pr = r;
      }
      aux[t] = first; ev_last[t] = last; fn[t] = g;
    });
    // the function of a chunk's first event needs the column + offset of the event before it: the offset-free part is the
    // previous event's raw column (take-right scan); the offset itself enters through the composition below
    block_scan(ex, ev_last, tmp, take_right);
    ex.phase([&](int t) {
      if (aux[t] != kNone) {
        // before the first event of the scan prev_col + shift = 0: the "previous column" is 0 with offset 0
        const int c_prev = t > 0 && ev_last[t - 1] != kNone ? ev_last[t - 1] : 0;
        fn[t] = fn_then(fn_of_columns(c_prev, aux[t]), fn[t]);
      }
    });
    block_scan(ex, fn, tmp, fn_then);
    ex.phase([&](int t) {                 // final walk of the chunk with the incoming offset
      int s = t > 0 ? fn_apply(fn[t - 1], 0) : 0;
      int c_prev = t > 0 && ev_last[t - 1] != kNone ? ev_last[t - 1] : 0;
      int pr = prev_ring_in[t], last_acc = kNone;
      for (int k = 0, i = t * L; k < L && i < n; ++k, ++i) {
        const int a = k * T + t;
        const PointRec q = rec[a];
        const int r = rec_ring(q);
        if (r < 0) { col_pos[2 * a] = -1; col_pos[2 * a + 1] = 0; continue; }
        int col = column_of(lo_rec(q), i >= W, sc.start_ori, w);
        if (firing_slot(r, rings) < firing_slot(pr, rings)) { s = fn_apply(fn_of_columns(c_prev, col), s); c_prev = col; }
        pr = r;
        col += s;
        while (col >= horizon) col -= horizon;
        col_pos[2 * a] = col < 0 ? -1 : col; col_pos[2 * a + 1] = 0;
        if (col >= 0) last_acc = i;
      }
      acc_last[t] = last_acc;
    });
    block_scan(ex, acc_last, tmp, take_right);
  };
  shift_pass(n);
  // the crossing test of every point against the last accepted point before it; aux = first point per chunk whose test is not "no"
  ex.phase([&](int t) {
    int prev = t > 0 && acc_last[t - 1] != kNone ? acc_last[t - 1] : -1;
    int found = 0x7fffffff, kind = 0, against = -1;
    for (int k = 0, i = t * L; k < L && i < n && found == 0x7fffffff; ++k, ++i) {
      const PointRec q = rec[k * T + t];
      if (rec_ring(q) < 0) continue;
      if (prev >= 0) {
        const double last_lo = lo_of(prev), last_hi = hi_of(prev);
        if (lo_rec(q) < last_hi) {
          const bool sure = hi_rec(q) < last_lo;
          int sure_behind = 0, maybe_behind = 0, seen = 0;
          for (int j = i + 1; j < i + rings + 1 && j < n; ++j) {
            sure_behind += hi_of(j) < last_lo ? 1 : 0;
            maybe_behind += lo_of(j) < last_hi ? 1 : 0;
            ++seen;
            if (maybe_behind < seen) break;
          }
          if (sure && sure_behind >= rings) { found = i; kind = 1; }
          else if (maybe_behind >= rings) { found = i; kind = 2; against = prev; }
        }
      }
      if (col_pos[2 * (k * T + t)] >= 0) prev = i;
    }
    aux[t] = found; fn[t] = kind; ev_last[t] = against;
  });
  ex.phase([&](int t) { tmp[t] = aux[t]; });
  ex.phase([&](int t) { acc_last[t] = tmp[t]; });
  block_scan(ex, acc_last, tmp, take_min);
  const int W = acc_last[T - 1];
  if (W != 0x7fffffff) {
    // the chunk that owns W knows the outcome
    int kind = 0, against = -1;
    for (int t = 0; t < T; ++t) if (aux[t] == W) { kind = fn[t]; against = ev_last[t]; }
    if (kind == 2) { *last_point = against; return W; }
    ex.phase([&](int) {});
    shift_pass(W);
  }
  *last_point = -1;
  // position inside the ring: per-ring prefix count of the accepted points, kRingGroup rings per walk of the chunk (a walk is a chain of
  // dependent loads: with one ring per walk the 2 x N_SCANS walks were 80 % of the kernel)
  int* cnt = scratch + 6 * T;             // kRingGroup x T
  int* cnt_tmp = cnt + kRingGroup * T;    // kRingGroup x T
  for (int r0 = 0; r0 < rings; r0 += kRingGroup) {
    ex.phase([&](int t) {
      int c[kRingGroup];
#pragma unroll
      for (int k = 0; k < kRingGroup; ++k) c[k] = 0;
      for (int j = 0, i = t * L; j < L && i < n; ++j, ++i) {
        const int a = j * T + t;
        const int r = col_pos[2 * a] >= 0 ? rec_ring(rec[a]) - r0 : -1;
#pragma unroll
        for (int k = 0; k < kRingGroup; ++k) c[k] += r == k ? 1 : 0;
      }
#pragma unroll
      for (int k = 0; k < kRingGroup; ++k) cnt[k * T + t] = c[k];
    });
    for (int off = 1; off < T; off <<= 1) {       // kRingGroup inclusive sum scans in the same phases
      ex.phase([&](int t) {
#pragma unroll
        for (int k = 0; k < kRingGroup; ++k) cnt_tmp[k * T + t] = t >= off ? cnt[k * T + t - off] + cnt[k * T + t] : cnt[k * T + t];
      });
      ex.phase([&](int t) {
#pragma unroll
        for (int k = 0; k < kRingGroup; ++k) cnt[k * T + t] = cnt_tmp[k * T + t];
      });
    }
    ex.phase([&](int t) {
      int run[kRingGroup];
#pragma unroll
      for (int k = 0; k < kRingGroup; ++k) run[k] = t > 0 ? cnt[k * T + t - 1] : 0;
      for (int j = 0, i = t * L; j < L && i < n; ++j, ++i) {
        const int a = j * T + t;
        const int r = col_pos[2 * a] >= 0 ? rec_ring(rec[a]) - r0 : -1;
        int pos = 0;
#pragma unroll
        for (int k = 0; k < kRingGroup; ++k) { if (r == k) pos = run[k]; run[k] += r == k ? 1 : 0; }
        if (r >= 0 && r < kRingGroup) col_pos[2 * a + 1] = pos;
      }
      if (t == T - 1) {
#pragma unroll
        for (int k = 0; k < kRingGroup; ++k) if (r0 + k < rings) ring_count[r0 + k] = cnt[k * T + T - 1];
      }
    });
  }
  for (int r = rings; r < kMaxRings; ++r) ring_count[r] = 0;
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
