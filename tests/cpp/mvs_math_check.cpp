// Host-compiled check of the per-texel device bodies in panovlm_amd/csrc/pvlm_mvs_core.h: the same functions the HIP
// kernel k_mvs_conf calls, composed serially (one "lane" after the other, plain left-to-right sums) so that the
// per-texel arithmetic and the composition logic can be compared with the oracle on a machine without a GPU
// (tests/test_mvs_cpu.py).  TEST INFRASTRUCTURE ONLY — libpvlm.so has no host path.
#include <vector>

#define PVLM_HD
#define PVLM_ATOMIC_MIN_U32(ptr, v) (*(ptr) = *(ptr) < (v) ? *(ptr) : (v))
#define PVLM_ATOMIC_MIN_U64(ptr, v) (*(ptr) = *(ptr) < (v) ? *(ptr) : (v))
#include "../../panovlm_amd/csrc/pvlm_mvs_core.h"

extern "C" void chk_mvs_conf(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                             const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                             const float* const* nei_depth) {
  using namespace pvlm_mvs;
  std::vector<float> unit((size_t)rows * cols * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  const int n = num_texels(half_window, step);
  std::vector<float> w(n), t0(n), t1(n);
  for (int py = 0; py < rows; ++py)
    for (int px = 0; px < cols; ++px) {
      const size_t e = (size_t)py * cols + px;
      const float dep = depth[e];
      if (dep <= 0) continue;
      float c = -1.f, sq0 = 0.f;
      const bool inside = px >= half_window && py >= half_window && px < cols - half_window && py < rows - half_window;
      if (inside) {
        float wsum = 0.f;
        for (int k = 0; k < n; ++k) { patch_texel(ref_gray, cols, px, py, half_window, step, k, &w[k], &t0[k]); wsum += w[k]; }
        float mean = 0.f;
        for (int k = 0; k < n; ++k) { w[k] /= wsum; mean += w[k] * t0[k]; }
        for (int k = 0; k < n; ++k) { t0[k] -= mean; const float tmp = t0[k] * w[k]; sq0 += t0[k] * tmp; t0[k] = tmp; }
      }
      if (inside && sq0 > 0) {
        const float* u0 = &unit[3 * e];
        const float X0[3] = {u0[0] * dep, u0[1] * dep, u0[2] * dep};
        const float* nr = normal + 3 * e;
        const float d = X0[0] * nr[0] + X0[1] * nr[1] + X0[2] * nr[2];
        if (!(d > 0)) {
          float best1 = 0.f, best2 = 0.f; int count = 0;
          for (int b = 0; b < n_neighbors; ++b) {
            float H[9];
            homography(R_nr + 9 * b, t_nr + 3 * b, nr, d, H);
            bool ok = true;
            for (int k = 0; k < n && ok; ++k) ok = neighbour_texel(unit.data(), nei_gray[b], rows, cols, H, px, py, half_window, step, k, &t1[k]);
            if (!ok) continue;
            float sum = 0.f, sq1 = 0.f, sq01 = 0.f;
            for (int k = 0; k < n; ++k) sum += t1[k] * w[k];
            for (int k = 0; k < n; ++k) t1[k] -= sum;
            for (int k = 0; k < n; ++k) sq1 += t1[k] * t1[k] * w[k];
            for (int k = 0; k < n; ++k) sq01 += t0[k] * t1[k];
            const float nrm = sq0 * sq1;
            if (nrm <= 0.f) continue;
            float score = sq01 / sqrtf(nrm);
            score = fminf(fmaxf(score, -1.f), 1.f);
            if (nei_depth) score = geometric_adjust(score, rows, cols, X0, R_nr + 9 * b, t_nr + 3 * b, nei_depth[b]);
            if (count == 0 || score > best1) { best2 = best1; best1 = score; } else if (count == 1 || score > best2) best2 = score;
            ++count;
          }
          if (count == 1) c = best1;
          else if (count >= 2) { float avg = 0.f; avg += best1; avg += best2; c = avg / 2; }
        }
      }
      conf[e] = c;
      if (c <= -1) { depth[e] = 0; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0; }
    }
}

extern "C" void chk_mvs_filter(int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* R_nr, const float* t_nr, const float* depth,
                               const float* conf, const unsigned char* depth_constant, float thr, float* depth_filter, float* conf_filter) {
  using namespace pvlm_mvs;
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(npix * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  std::vector<unsigned> proj(npix * (size_t)(n_neighbors > 0 ? n_neighbors : 1), 0x7f800000u);
  for (int b = 0; b < n_neighbors; ++b) {
    float R_rn[9], t_rn[3];
    inverse_pose(R_nr + 9 * b, t_nr + 3 * b, R_rn, t_rn);
    for (size_t e = 0; e < npix; ++e) project_splat(rows, cols, unit.data(), nei_depth[b], R_rn, t_rn, (long long)e, proj.data() + npix * b);
  }
  for (size_t e = 0; e < npix; ++e) filter_pixel(rows, cols, n_neighbors, proj.data(), depth, conf, depth_constant, thr, (long long)e, depth_filter, conf_filter);
}

// FilterDepthImageRefine through the device bodies.  The source pixels are visited in REVERSE raster order on purpose:
// the keyed minimum must reproduce the reference's sequential "last writer among the closest" whatever the order.
extern "C" void chk_mvs_filter_refine(int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* const* nei_conf, const float* R_nr,
                                      const float* t_nr, const float* depth, float* conf, const unsigned char* depth_constant, float thr, float min_depth,
                                      float max_depth, float* depth_filter, float* conf_filter) {
  using namespace pvlm_mvs;
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(npix * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  std::vector<unsigned long long> key(npix * (size_t)(n_neighbors > 0 ? n_neighbors : 1), ~0ull);
  RefineViews nv;
  nv.n = n_neighbors;
  for (int b = 0; b < n_neighbors; ++b) {
    float R_rn[9], t_rn[3];
    inverse_pose(R_nr + 9 * b, t_nr + 3 * b, R_rn, t_rn);
    for (size_t e = npix; e-- > 0;) project_splat_conf(rows, cols, unit.data(), nei_depth[b], R_rn, t_rn, (long long)e, key.data() + npix * b);
    nv.conf[b] = nei_conf[b];
    for (int k = 0; k < 9; ++k) nv.R[b][k] = R_nr[9 * b + k];
    for (int k = 0; k < 3; ++k) nv.t[b][k] = t_nr[3 * b + k];
  }
  for (size_t e = 0; e < npix; ++e)
    refine_pixel(rows, cols, nv, key.data(), unit.data(), depth, conf, depth_constant, thr, min_depth, max_depth, (long long)e, depth_filter, conf_filter);
}

// ---- PatchMatch sweep through the device bodies (process_pixel and everything it calls), serial scorer ----
namespace {
struct SerialPatch { std::vector<float> w, t0; float sq0 = 0.f; bool inside = false; };
void serial_fill_patch(const unsigned char* ref_gray, int rows, int cols, int px, int py, int half_window, int step, SerialPatch& P) {
  using namespace pvlm_mvs;
  const int n = num_texels(half_window, step);
  P.w.assign(n, 0.f); P.t0.assign(n, 0.f); P.sq0 = 0.f;
  P.inside = px >= half_window && py >= half_window && px < cols - half_window && py < rows - half_window;
  if (!P.inside) return;
  float wsum = 0.f;
  for (int k = 0; k < n; ++k) { patch_texel(ref_gray, cols, px, py, half_window, step, k, &P.w[k], &P.t0[k]); wsum += P.w[k]; }
  float mean = 0.f;
  for (int k = 0; k < n; ++k) { P.w[k] /= wsum; mean += P.w[k] * P.t0[k]; }
  for (int k = 0; k < n; ++k) { P.t0[k] -= mean; const float tmp = P.t0[k] * P.w[k]; P.sq0 += P.t0[k] * tmp; P.t0[k] = tmp; }
}
struct SerialScorer : pvlm_mvs::SerialMath, pvlm_mvs::SerialFactors {
  int rows, cols, half_window, step, px, py, n_neighbors;
  const float* unit; const unsigned char* const* nei_gray; const float* R_nr; const float* t_nr; const float* const* nei_depth; const SerialPatch* P;
  float operator()(const float* nr, float dep, const float* factors, int n_close) const {
    using namespace pvlm_mvs;
    const int n = num_texels(half_window, step);
    const float* u0 = unit + 3 * ((size_t)py * cols + px);
    const float X0[3] = {u0[0] * dep, u0[1] * dep, u0[2] * dep};
    const float d = X0[0] * nr[0] + X0[1] * nr[1] + X0[2] * nr[2];
    if (d > 0) return -1.f;
    std::vector<float> t1(n);
    float best1 = 0.f, best2 = 0.f; int count = 0;
    for (int b = 0; b < n_neighbors; ++b) {
      float H[9];
      homography(R_nr + 9 * b, t_nr + 3 * b, nr, d, H);
      bool ok = true;
      for (int k = 0; k < n && ok; ++k) ok = neighbour_texel(unit, nei_gray[b], rows, cols, H, px, py, half_window, step, k, &t1[k]);
      if (!ok) continue;
      float sum = 0.f, sq1 = 0.f, sq01 = 0.f;
      for (int k = 0; k < n; ++k) sum += t1[k] * P->w[k];
      for (int k = 0; k < n; ++k) t1[k] -= sum;
      for (int k = 0; k < n; ++k) sq1 += t1[k] * t1[k] * P->w[k];
      for (int k = 0; k < n; ++k) sq01 += P->t0[k] * t1[k];
      const float nrm = P->sq0 * sq1;
      if (nrm <= 0.f) continue;
      float score = sq01 / sqrtf(nrm);
      score = fminf(fmaxf(score, -1.f), 1.f);
      score = smooth_score(score, factors, n_close);
      if (nei_depth) score = geometric_adjust(score, rows, cols, X0, R_nr + 9 * b, t_nr + 3 * b, nei_depth[b]);
      if (count == 0 || score > best1) { best2 = best1; best1 = score; } else if (count == 1 || score > best2) best2 = score;
      ++count;
    }
    if (count == 1) return best1;
    if (count >= 2) { float avg = 0.f; avg += best1; avg += best2; return avg / 2; }
    return -1.f;
  }
};
}  // namespace

extern "C" void chk_mvs_propagate(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                  const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                  const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                  unsigned long long seed, int max_iter, float conf_threshold) {
  using namespace pvlm_mvs;
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(npix * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  SerialPatch P;
  for (int iter = 0; iter < max_iter; ++iter)
    for (int offset = 0; offset <= 1; ++offset) {
      const unsigned long long ps = pass_seed(seed, 2 * iter + offset);
      // the same wave -> pixel mapping as k_mvs_propagate, walked backwards: a colour pass must not depend on the order
      const int half = (cols + 1) / 2;
      for (long long wv = (long long)rows * half - 1; wv >= 0; --wv) {
        const int py = (int)(wv / half), px = ((py % 2 + offset) % 2) + 2 * (int)(wv % half);
        if (px >= cols) continue;
        const size_t e = (size_t)py * cols + px;
        float dep = depth[e];
        if (dep <= 0) continue;
        serial_fill_patch(ref_gray, rows, cols, px, py, half_window, step, P);
        if (!P.inside || P.sq0 <= 1e-6) continue;
        float nr[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
        float c = conf[e];
        SweepArgs A{rows, cols, unit.data(), depth, normal, depth_constant, min_depth, max_depth};
        Rng rng{ps, (unsigned long long)e, 0u};
        SerialScorer scorer{{}, {}, rows, cols, half_window, step, px, py, n_neighbors, unit.data(), nei_gray, R_nr, t_nr, nei_depth, &P};
        const int pdx[4] = {-1, 0, 1, 0}, pdy[4] = {0, -1, 0, 1};
        process_pixel(A, rng, px, py, scorer, dep, nr, c, 4, pdx, pdy);
        depth[e] = dep; normal[3 * e] = nr[0]; normal[3 * e + 1] = nr[1]; normal[3 * e + 2] = nr[2]; conf[e] = c;
      }
    }
  for (size_t e = 0; e < npix; ++e) {
    if (depth_constant && depth_constant[e]) continue;
    if (conf[e] < conf_threshold) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0.f; }
  }
}

// The checkerboard sweep through the THREAD-per-pixel bodies (pvlm_mvs::fill_patch_column + ColumnScorer, the program of
// k_mvs_propagate_lane): weight table and texel column with the strides the kernel uses ([texel][pixel of the pass], [texel][64]),
// pixels visited backwards.
namespace {
struct HostViews { const unsigned char* gray[16]; const float* depth[16]; float R[16][9]; float t[16][3]; int n; int geometric;
                   const unsigned char* image(int b) const { return gray[b]; } };
}
extern "C" void chk_mvs_propagate_column(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                         const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                         const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                         unsigned long long seed, int max_iter, float conf_threshold) {
  using namespace pvlm_mvs;
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(npix * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  HostViews nb{};
  nb.n = n_neighbors; nb.geometric = nei_depth != nullptr;
  for (int b = 0; b < n_neighbors; ++b) {
    nb.gray[b] = nei_gray[b]; nb.depth[b] = nei_depth ? nei_depth[b] : nullptr;
    for (int k = 0; k < 9; ++k) nb.R[b][k] = R_nr[9 * b + k];
    for (int k = 0; k < 3; ++k) nb.t[b][k] = t_nr[3 * b + k];
  }
  const int n = num_texels(half_window, step), half = (cols + 1) / 2;
  const size_t in_pass = (size_t)rows * half;
  std::vector<float> wtab((size_t)n * in_pass), t1((size_t)n * 64);
  for (int iter = 0; iter < max_iter; ++iter)
    for (int offset = 0; offset <= 1; ++offset) {
      const unsigned long long ps = pass_seed(seed, 2 * iter + offset);
      for (long long wv = (long long)in_pass - 1; wv >= 0; --wv) {
        const int py = (int)(wv / half), px = ((py % 2 + offset) % 2) + 2 * (int)(wv % half);
        if (px >= cols) continue;
        const size_t e = (size_t)py * cols + px;
        float dep = depth[e];
        if (dep <= 0) continue;
        ColumnPatch P{wtab.data() + wv, in_pass, t1.data() + (wv % 64), 64, 0.f, 0.f, false};
        fill_patch_column(ref_gray, rows, cols, px, py, half_window, step, n, P);
        if (!P.inside || P.sq0 <= 1e-6) continue;
        float nr[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
        float c = conf[e];
        SweepArgs A{rows, cols, unit.data(), depth, normal, depth_constant, min_depth, max_depth};
        Rng rng{ps, (unsigned long long)e, 0u};
        ColumnScorer<HostViews> scorer{{}, {}, rows, cols, half_window, step, n, px, py, unit.data(), ref_gray, &nb, P};
        const int pdx[4] = {-1, 0, 1, 0}, pdy[4] = {0, -1, 0, 1};
        process_pixel(A, rng, px, py, scorer, dep, nr, c, 4, pdx, pdy);
        depth[e] = dep; normal[3 * e] = nr[0]; normal[3 * e + 1] = nr[1]; normal[3 * e + 2] = nr[2]; conf[e] = c;
      }
    }
  for (size_t e = 0; e < npix; ++e) {
    if (depth_constant && depth_constant[e]) continue;
    if (conf[e] < conf_threshold) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0.f; }
  }
}

// The sequential sweep through the device bodies, in the order k_mvs_propagate_diag gives the GPU: anti-diagonal after
// anti-diagonal, and INSIDE a diagonal from the bottom row up (the launch makes no promise about the order of its waves) — not
// the raster order of the oracle.  Equal maps prove what the kernel relies on: the pixels of a diagonal do not depend on each other.
static void propagate_sequential_impl(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                      const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                      const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                      unsigned long long seed, int max_iter, float conf_threshold, int spec_width) {
  using namespace pvlm_mvs;
  const size_t npix = (size_t)rows * cols;
  std::vector<float> unit(npix * 3);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) unit_ray(rows, cols, c, r, &unit[3 * ((size_t)r * cols + c)]);
  SerialPatch P;
  const int n_diag = rows + cols - 1;
  for (int iter = 0; iter < max_iter; ++iter) {
    const unsigned long long ps = pass_seed(seed, iter);
    const int backward = iter % 2, sgn = backward ? 1 : -1;
    for (int q = 0; q < n_diag; ++q) {
      const int d = backward ? n_diag - 1 - q : q;
      const int r0 = std::max(0, d - (cols - 1)), r1 = std::min(rows - 1, d);
      for (int py = r1; py >= r0; --py) {
        const int px = d - py;
        const size_t e = (size_t)py * cols + px;
        float dep = depth[e];
        if (dep <= 0) continue;
        serial_fill_patch(ref_gray, rows, cols, px, py, half_window, step, P);
        if (!P.inside || P.sq0 <= 0) continue;
        float nr[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
        float c = conf[e];
        SweepArgs A{rows, cols, unit.data(), depth, normal, depth_constant, min_depth, max_depth};
        Rng rng{ps, (unsigned long long)e, 0u};
        SerialScorer scorer{{}, {}, rows, cols, half_window, step, px, py, n_neighbors, unit.data(), nei_gray, R_nr, t_nr, nei_depth, &P};
        const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
        if (spec_width > 0) {          // the speculative batches of k_mvs_propagate_diag_spec, the "waves" of a batch one after the other
          SerialBatch<SerialScorer> batch{&scorer, spec_width};
          process_pixel_spec(A, rng, px, py, batch, dep, nr, c, 2, pdx, pdy);
        } else {
          process_pixel(A, rng, px, py, scorer, dep, nr, c, 2, pdx, pdy);
        }
        depth[e] = dep; normal[3 * e] = nr[0]; normal[3 * e + 1] = nr[1]; normal[3 * e + 2] = nr[2]; conf[e] = c;
      }
    }
  }
  for (size_t e = 0; e < npix; ++e) {
    if (depth_constant && depth_constant[e]) continue;
    if (conf[e] < conf_threshold) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = normal[3 * e + 1] = normal[3 * e + 2] = 0.f; }
  }
}
extern "C" void chk_mvs_propagate_sequential(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                             const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                             const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                             unsigned long long seed, int max_iter, float conf_threshold) {
  propagate_sequential_impl(rows, cols, half_window, step, ref_gray, n_neighbors, nei_gray, R_nr, t_nr, depth, normal, conf, nei_depth, depth_constant, min_depth, max_depth,
                            seed, max_iter, conf_threshold, 0);
}
// the same sweep through process_pixel_spec: batches of `width` hypotheses scored side by side (width = 4 on the GPU)
extern "C" void chk_mvs_propagate_sequential_spec(int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                                  const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                                  const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                                  unsigned long long seed, int max_iter, float conf_threshold, int width) {
  propagate_sequential_impl(rows, cols, half_window, step, ref_gray, n_neighbors, nei_gray, R_nr, t_nr, depth, normal, conf, nei_depth, depth_constant, min_depth, max_depth,
                            seed, max_iter, conf_threshold, width);
}
// the per-pixel bodies of k_cloud_count / k_cloud_emit (pvlm_mvs.hip) run on the host in raster order
extern "C" long long chk_mvs_depth_to_cloud(int rows, int cols, const float* depth, const unsigned char* bgr, const float* normal, const double* T_wc,
                                            float max_depth, int filter_sky, float* xyz, unsigned char* rgb, float* normal_out) {
  long long n = 0;
  for (long long e = 0; e < (long long)rows * cols; ++e) {
    if (!pvlm_mvs::cloud_keeps(depth[e], max_depth, bgr + 3 * e, filter_sky != 0)) continue;
    float ray[3];
    pvlm_mvs::unit_ray(rows, cols, (int)(e % cols), (int)(e / cols), ray);
    pvlm_mvs::cloud_point(ray, depth[e], T_wc, xyz + 3 * n);
    rgb[3 * n] = bgr[3 * e + 2]; rgb[3 * n + 1] = bgr[3 * e + 1]; rgb[3 * n + 2] = bgr[3 * e];
    if (normal_out) pvlm_mvs::cloud_normal(normal + 3 * e, T_wc, normal_out + 3 * n);
    ++n;
  }
  return n;
}

extern "C" unsigned chk_mvs_random_u32(unsigned long long seed, unsigned long long pixel, unsigned k) { return pvlm_mvs::random_u32(seed, pixel, k); }
