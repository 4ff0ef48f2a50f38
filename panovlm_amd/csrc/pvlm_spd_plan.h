// Ordering + symbolic factorisation of the tile-sparse pose solve (pvlm_spd_solve_blocks, csrc/pvlm_linalg.hip) — pure host code, so that
// tests/cpp/spd_plan_check.cpp can check it without a GPU (tests/test_spd_plan_cpu.py: the lists cover every nonzero of a numeric factor).
//   1. nodes = groups of unknowns that always appear together (the 6 parameters of a pose); minimum-degree ordering of the node graph
//      (elimination game on bitset rows), then the postorder of its elimination tree, so that nodes with the same structure sit next to
//      each other (minimum degree alone scatters them: 93 % of the dense factorisation's tile updates remain; with the postorder 21 %);
//   2. symbolic factorisation at the granularity the kernels work at: which 64-row tiles hold a nonzero of which NB-column block column,
//      fill included;
//   3. per block column the list of its row tiles (panel) and of the tile pairs its rank-NB update touches.
#pragma once
#include <algorithm>
#include <vector>

namespace pvlm_spd {

struct TilePair { int row_tile, col_tile; };

struct Symbolic {
  std::vector<int> new_of_old;                 // permutation of the unknowns
  std::vector<int> row_tiles;                  // concatenated over the block columns
  std::vector<TilePair> pairs;
  std::vector<int> row_off, pair_off;          // block columns + 1
  double update_fraction = 1.0;                // tile updates / tile updates of the dense factorisation
  bool ordered = false;                        // false: the natural order was kept (too many nodes for the bitset ordering)
};

// row_idx / col_idx: 6 scalar indices per block side (-1 or >= n: constant, dropped).  NB: block column width of the kernels (32).
inline void plan_symbolic(int n, int n_blocks, const int* row_idx, const int* col_idx, int NB, Symbolic* P) {
  P->new_of_old.resize((size_t)n);
  for (int i = 0; i < n; ++i) P->new_of_old[(size_t)i] = i;
  P->row_tiles.clear(); P->pairs.clear(); P->row_off.assign(1, 0); P->pair_off.assign(1, 0); P->update_fraction = 1.0; P->ordered = false;
  if (n <= 0) return;
  // ---- nodes: unknowns that share a block side
  std::vector<int> uf((size_t)n);
  for (int i = 0; i < n; ++i) uf[(size_t)i] = i;
  auto find = [&](int a) { while (uf[(size_t)a] != a) { uf[(size_t)a] = uf[(size_t)uf[(size_t)a]]; a = uf[(size_t)a]; } return a; };
  auto side = [&](const int* idx) { int first = -1; for (int r = 0; r < 6; ++r) { const int i = idx[r]; if (i < 0 || i >= n) continue; if (first < 0) first = find(i); else { const int q = find(i); if (q != first) uf[(size_t)std::max(q, first)] = std::min(q, first), first = std::min(q, first); } } return first; };
  for (int b = 0; b < n_blocks; ++b) { side(row_idx + 6 * b); side(col_idx + 6 * b); }
  std::vector<int> node_of((size_t)n, -1);
  std::vector<std::vector<int>> members;
  for (int i = 0; i < n; ++i) { const int r = find(i); if (node_of[(size_t)r] < 0) { node_of[(size_t)r] = (int)members.size(); members.emplace_back(); } node_of[(size_t)i] = node_of[(size_t)r]; members[(size_t)node_of[(size_t)i]].push_back(i); }
  const int N = (int)members.size();
  std::vector<std::vector<int>> adj((size_t)N);
  for (int b = 0; b < n_blocks; ++b) {
    int a = -1, c = -1;
    for (int r = 0; r < 6 && a < 0; ++r) if (row_idx[6 * b + r] >= 0 && row_idx[6 * b + r] < n) a = node_of[(size_t)row_idx[6 * b + r]];
    for (int r = 0; r < 6 && c < 0; ++r) if (col_idx[6 * b + r] >= 0 && col_idx[6 * b + r] < n) c = node_of[(size_t)col_idx[6 * b + r]];
    if (a >= 0 && c >= 0 && a != c) { adj[(size_t)a].push_back(c); adj[(size_t)c].push_back(a); }
  }
  for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  // ---- minimum degree: the elimination game on bitset rows (N is a few thousand: eliminating a node ORs its row into its neighbours',
  // 25 words each at Floor size; sorted-list merges took 70 ms per plan there, this takes 3)
  if (N > 16384) return;
  std::vector<int> order; order.reserve((size_t)N);
  {
    const int W = (N + 63) / 64;
    std::vector<unsigned long long> bits((size_t)N * W, 0ull);
    std::vector<int> degree((size_t)N, 0);
    for (int v = 0; v < N; ++v) { for (int a : adj[(size_t)v]) bits[(size_t)v * W + (a >> 6)] |= 1ull << (a & 63); degree[(size_t)v] = (int)adj[(size_t)v].size(); }
    std::vector<char> gone((size_t)N, 0);
    std::vector<int> nb;
    for (int step = 0; step < N; ++step) {
      int best = -1, deg = 0x7fffffff;
      for (int v = 0; v < N; ++v) if (!gone[(size_t)v] && degree[(size_t)v] < deg) { deg = degree[(size_t)v]; best = v; }
      gone[(size_t)best] = 1; order.push_back(best);
      const unsigned long long* rb = &bits[(size_t)best * W];
      nb.clear();
      for (int w = 0; w < W; ++w) for (unsigned long long m = rb[w]; m; m &= m - 1) nb.push_back(w * 64 + __builtin_ctzll(m));
      for (int a : nb) {
        unsigned long long* ra = &bits[(size_t)a * W];
        int d = 0;
        for (int w = 0; w < W; ++w) { ra[w] |= rb[w]; }
        ra[a >> 6] &= ~(1ull << (a & 63)); ra[best >> 6] &= ~(1ull << (best & 63));
        for (int w = 0; w < W; ++w) d += __builtin_popcountll(ra[w]);
        degree[(size_t)a] = d;
      }
    }
  }
  // ---- elimination tree of the permuted graph + postorder
  {
    std::vector<int> pos((size_t)N);
    for (int k = 0; k < N; ++k) pos[(size_t)order[(size_t)k]] = k;
    std::vector<int> parent((size_t)N, -1), anc((size_t)N, -1);
    for (int i = 0; i < N; ++i)
      for (int w : adj[(size_t)order[(size_t)i]]) {
        int j = pos[(size_t)w];
        while (j != -1 && j < i) { const int nxt = anc[(size_t)j]; anc[(size_t)j] = i; if (nxt == -1) parent[(size_t)j] = i; j = nxt; }
      }
    std::vector<std::vector<int>> child((size_t)N);
    std::vector<int> roots;
    for (int v = 0; v < N; ++v) { if (parent[(size_t)v] >= 0) child[(size_t)parent[(size_t)v]].push_back(v); else roots.push_back(v); }
    std::vector<int> post; post.reserve((size_t)N);
    std::vector<std::pair<int, size_t>> stack;
    for (int r : roots) {
      stack.push_back({r, 0});
      while (!stack.empty()) {
        auto& top = stack.back();
        if (top.second < child[(size_t)top.first].size()) { const int c = child[(size_t)top.first][top.second++]; stack.push_back({c, 0}); }
        else { post.push_back(top.first); stack.pop_back(); }
      }
    }
    std::vector<int> reordered((size_t)N);
    for (int k = 0; k < N; ++k) reordered[(size_t)k] = order[(size_t)post[(size_t)k]];
    order.swap(reordered);
  }
  {
    int next = 0;
    for (int v : order) for (int i : members[(size_t)v]) P->new_of_old[(size_t)i] = next++;
  }
  // ---- symbolic factorisation on (64-row tile) x (32-column block column) cells
  const int C = (n + NB - 1) / NB, T = (n + 63) / 64;
  std::vector<unsigned char> nz((size_t)T * C, 0);
  for (int b = 0; b < n_blocks; ++b)
    for (int r = 0; r < 6; ++r) {
      const int io = row_idx[6 * b + r];
      if (io < 0 || io >= n) continue;
      const int i = P->new_of_old[(size_t)io];
      for (int c = 0; c < 6; ++c) {
        const int jo = col_idx[6 * b + c];
        if (jo < 0 || jo >= n) continue;
        const int j = P->new_of_old[(size_t)jo];
        const int lo = std::min(i, j), hi = std::max(i, j);
        nz[(size_t)(hi / 64) * C + lo / NB] = 1;
      }
    }
  std::vector<int>& row_tiles = P->row_tiles; std::vector<TilePair>& pairs = P->pairs;
  P->ordered = true;
  long long dense_pairs = 0;
  std::vector<int> R;
  for (int k = 0; k < C; ++k) {
    const int base = std::min(n, (k + 1) * NB);
    const int t0 = base / 64;
    R.clear();
    for (int t = t0; t < T && base < n; ++t) if (nz[(size_t)t * C + k] && (t + 1) * 64 > base) R.push_back(t);   // tiles with a row below the diagonal block
    for (int t : R) row_tiles.push_back(t);
    for (size_t a = 0; a < R.size(); ++a)
      for (size_t c = 0; c <= a; ++c) {
        pairs.push_back(TilePair{R[a], R[c]});
        // the block columns a 64-row tile spans: 64 / NB of them (two at the library's NB = 32)
        for (int col = (64 / NB) * R[c]; col < (64 / NB) * (R[c] + 1); ++col) if (col > k && col < C) nz[(size_t)R[a] * C + col] = 1;
      }
    P->row_off.push_back((int)row_tiles.size()); P->pair_off.push_back((int)pairs.size());
    const long long dt = (n - base + 63) / 64;
    dense_pairs += dt * (dt + 1) / 2;
  }
  P->update_fraction = dense_pairs ? (double)pairs.size() / (double)dense_pairs : 1.0;
}

}  // namespace pvlm_spd
