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
#ifdef PVLM_PLAN_PROFILE      // section timers of plan_levels on stdout (development)
#include <chrono>
#include <cstdio>
#endif
#include <cstdlib>
#include <utility>
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


// ---- the factorisation as ONE launch (k_nd_flow): tasks handed from workgroup to workgroup inside the launch --------------------------------------------------------
// The level schedule still costs three dependent launches per level (46 levels + the tail on the Floor graph, ~90 us each: 4.2 of the 6.2 ms of a solve), and a level is
// as slow as its slowest launch ramp although most of its tiles were ready long before.  plan_flow restates the same symbolic factorisation at 64 x 64 TILE granularity
// as a list of TASKS: task (I, J) = tile row I of tile column J; it subtracts L(I,K) L(J,K)^T for every earlier tile column K that holds both tiles (its sources, in
// ascending elimination order), then factorises (I == J) or multiplies by the inverse of the diagonal tile (I > J).  Tasks are ordered by the dependency depth of
// their column, then column, diagonal first — every task depends only on tasks before it, so workgroups that take tasks in this order by a ticket can always finish
// the lowest unfinished one (no residency assumption).  The backward substitution walks the columns in reverse order.
struct FlowTask { int I, J, src_off, n_src, prev, final_, pad0, pad1; };   // prev: the task that must have published this tile before it is read (-1: none);
                                                                            // final_ = 0: a CHUNK — it only subtracts its sources from the tile in memory
struct FlowSource { int K, task_a, task_b, pad; };           // tile column K; the tasks that publish L(I,K) and L(J,K)
struct FlowPlan {
  int tile_cols = 0, depth = 0, finals = 0;                   // 64-wide tile columns; dependent tile columns on the longest chain; tasks that finish a tile
  std::vector<FlowTask> tasks;
  std::vector<FlowSource> sources;
  std::vector<int> col_order;                                 // tile columns in task order
  std::vector<int> below_off, below;                          // per tile column (tile_cols + 1, indexed by COLUMN): its tile rows below the diagonal, ascending
  bool ready = false;
};

// row_off / row_tiles: the panel tiles of every NB-column block column, as plan_levels leaves them (tiles that reach below the block, the block's own tile included
// for the first half of a tile column).  chunk: sources per chunk task (0: no chunks).
// CHUNKS.  A tile of the top separator has a source in nearly every column below it (138 on the Floor graph): one task adding them up one after the other is 0.4 ms of
// serial work that starts when the task gets its ticket — late, tickets go in dependency order — and the tile column that needs it waits (measured: 170 us between
// a column's inverse and the first dependency of the next).  So the sources of a tile that are published two or more levels before the tile's own column are split off
// into chunk tasks of `chunk` sources, each placed right after the level that publishes its last source: they run while the launch has idle workgroups, subtract their
// products from the tile IN MEMORY (one after the other: chunk c waits for chunk c - 1), and the final task is left with the sources of the level just before it.
inline void plan_flow(int n_pad, int NB, const std::vector<int>& row_off, const std::vector<int>& row_tiles, int chunk, FlowPlan* F) {
  *F = FlowPlan();
  const int per_tile = 64 / NB, TC = n_pad / 64;
  if (TC <= 0 || n_pad % 64 != 0 || (int)row_off.size() != TC * per_tile + 1) return;
  F->tile_cols = TC;
  F->below_off.assign(1, 0);
  std::vector<unsigned char> mark((size_t)TC, 0);
  for (int J = 0; J < TC; ++J) {
    const size_t first = F->below.size();
    for (int k = per_tile * J; k < per_tile * (J + 1); ++k)
      for (int q = row_off[(size_t)k]; q < row_off[(size_t)k + 1]; ++q) { const int t = row_tiles[(size_t)q]; if (t > J && t < TC && !mark[(size_t)t]) { mark[(size_t)t] = 1; F->below.push_back(t); } }
    std::sort(F->below.begin() + (long)first, F->below.end());
    for (size_t q = first; q < F->below.size(); ++q) mark[(size_t)F->below[q]] = 0;
    F->below_off.push_back((int)F->below.size());
  }
  // dependency depth of the tile columns, their order
  std::vector<int> lev((size_t)TC, 0);
  for (int K = 0; K < TC; ++K) for (int q = F->below_off[(size_t)K]; q < F->below_off[(size_t)K + 1]; ++q) { int& l = lev[(size_t)F->below[(size_t)q]]; l = std::max(l, lev[(size_t)K] + 1); }
  F->col_order.resize((size_t)TC);
  for (int J = 0; J < TC; ++J) { F->col_order[(size_t)J] = J; F->depth = std::max(F->depth, lev[(size_t)J] + 1); }
  std::stable_sort(F->col_order.begin(), F->col_order.end(), [&](int a, int b) { return lev[(size_t)a] < lev[(size_t)b]; });
  // the tiles ("finals": the tasks that finish a tile), column by column in that order; first tile of column J = its diagonal tile
  struct Tile { int I, J, src_off, n_src, n_chunks, task; };
  std::vector<Tile> tiles;
  std::vector<int> first_tile((size_t)TC, 0);
  for (int J : F->col_order) {
    first_tile[(size_t)J] = (int)tiles.size();
    tiles.push_back(Tile{J, J, 0, 0, 0, -1});
    for (int q = F->below_off[(size_t)J]; q < F->below_off[(size_t)J + 1]; ++q) tiles.push_back(Tile{F->below[(size_t)q], J, 0, 0, 0, -1});
  }
  F->finals = (int)tiles.size();
  auto tile_of = [&](int I, int J) {
    if (I == J) return first_tile[(size_t)J];
    const int* b = F->below.data() + F->below_off[(size_t)J]; const int* e = F->below.data() + F->below_off[(size_t)J + 1];
    const int* it = std::lower_bound(b, e, I);
    return (it != e && *it == I) ? first_tile[(size_t)J] + 1 + (int)(it - b) : -1;
  };
  // sources per tile as (K, tile of (I,K), tile of (J,K)): two passes over the columns IN TASK ORDER (count, fill) — inside a tile ascending (level, column) of K
  struct Src { int K, ta, tb; };
  std::vector<Src> srcs;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) { int off = 0; for (Tile& t : tiles) { t.src_off = off; off += t.n_src; t.n_src = 0; } srcs.assign((size_t)off, Src{0, 0, 0}); }
    for (int K : F->col_order) {
      const int* b = F->below.data() + F->below_off[(size_t)K]; const int nb = F->below_off[(size_t)K + 1] - F->below_off[(size_t)K];
      for (int c = 0; c < nb; ++c)
        for (int a = c; a < nb; ++a) {
          const int id = tile_of(b[a], b[c]);
          if (id < 0) return;                                   // a fill the symbolic factorisation did not mark: not ready (never seen; checked by the CPU test)
          Tile& t = tiles[(size_t)id];
          if (pass == 1) srcs[(size_t)(t.src_off + t.n_src)] = Src{K, first_tile[(size_t)K] + 1 + a, first_tile[(size_t)K] + 1 + c};
          ++t.n_src;
        }
    }
  }
  // chunks: the sources published two levels or more before the tile's column, `chunk` at a time (the last chunk takes the remainder when it is short)
  struct Chunk { int tile, first, count, ready_level; };
  std::vector<std::vector<Chunk>> chunks_at((size_t)F->depth);
  for (size_t f = 0; f < tiles.size(); ++f) {
    Tile& t = tiles[f];
    int early = 0;
    while (chunk > 0 && early < t.n_src && lev[(size_t)srcs[(size_t)(t.src_off + early)].K] + 2 <= lev[(size_t)t.J]) ++early;
    if (chunk <= 0 || early < std::max(4, chunk / 2)) continue;
    const int nc = (early + chunk - 1) / chunk;
    for (int c = 0; c < nc; ++c) {
      const int first = (int)((long long)early * c / nc), last = (int)((long long)early * (c + 1) / nc);
      chunks_at[(size_t)lev[(size_t)srcs[(size_t)(t.src_off + last - 1)].K]].push_back(Chunk{(int)f, first, last - first, 0});
    }
    t.n_chunks = nc;
  }
  // task order: level by level — the tiles of the level's columns, then the chunks whose last source that level publishes
  std::vector<int> last_task_of_tile(tiles.size(), -1);          // the latest chunk of the tile emitted so far
  std::vector<int> chunk_first(tiles.size(), 0);                // sources of the tile already given to chunks
  size_t tile_cursor = 0;
  struct Pending { int task, tile, first, count; };
  std::vector<Pending> emit;
  for (int l = 0; l < F->depth; ++l) {
    while (tile_cursor < tiles.size() && lev[(size_t)tiles[tile_cursor].J] == l) {
      Tile& t = tiles[tile_cursor];
      t.task = (int)F->tasks.size();
      F->tasks.push_back(FlowTask{t.I, t.J, 0, t.n_src - chunk_first[tile_cursor], last_task_of_tile[tile_cursor], 1, 0, 0});
      emit.push_back(Pending{t.task, (int)tile_cursor, chunk_first[tile_cursor], t.n_src - chunk_first[tile_cursor]});
      ++tile_cursor;
    }
    for (const Chunk& c : chunks_at[(size_t)l]) {
      const Tile& t = tiles[(size_t)c.tile];
      const int id = (int)F->tasks.size();
      F->tasks.push_back(FlowTask{t.I, t.J, 0, c.count, last_task_of_tile[(size_t)c.tile], 0, 0, 0});
      emit.push_back(Pending{id, c.tile, c.first, c.count});
      last_task_of_tile[(size_t)c.tile] = id;
      chunk_first[(size_t)c.tile] = c.first + c.count;
    }
  }
  if (tile_cursor != tiles.size()) return;
  int off = 0;
  for (const Pending& e : emit) {
    FlowTask& t = F->tasks[(size_t)e.task];
    t.src_off = off;
    for (int q = 0; q < e.count; ++q) { const Src& sr = srcs[(size_t)(tiles[(size_t)e.tile].src_off + e.first + q)]; F->sources.push_back(FlowSource{sr.K, tiles[(size_t)sr.ta].task, tiles[(size_t)sr.tb].task, 0}); }
    off += e.count;
  }
  F->ready = true;
}

// ---- nested dissection + level schedule (round 6) -----------------------------------------------------------------------------------------------
// plan_symbolic above orders by minimum degree and factorises block column after block column: on the Floor pose graph (1 593 poses, every pose tied to
// ~20 others all along the trajectory — the scans of a room seen again and again) its 299 block columns form a dependency chain of 293 (every 64-row
// tile drags its neighbours along), 35 us each.  plan_levels orders by NESTED DISSECTION of the pose graph — recursive bisection by a breadth-first level
// set from a pseudo-peripheral node (George's automatic nested dissection); the two halves first, the separator last — aligns every group (leaf or
// separator) to a 64-row tile with dummy unknowns (identity rows), so that no tile couples two groups that have nothing to do with each other, and
// then SCHEDULES: the level of a block column = 1 + the highest level among the block columns whose update reaches it.  Block columns of one level are
// factorised in one launch, their trailing updates are applied in one launch in which a workgroup owns a TARGET tile and adds up every source column of the
// level in list order (deterministic: no two workgroups write the same tile, no atomics), the triangular solves run level by level too.  Floor graph:
// 110 levels instead of 293 dependent steps.
struct Target { int ti, tj, src_off, n_src, col_min, pad; };   // tile pair (ti >= tj, absolute 64-row tiles) <- sources[src_off .. + n_src) (block columns); columns below
                                                               // col_min are not written: the panel of a source column that lies inside the target's own column tile
struct RowTarget { int tile, src_off, n_src, pad; };    // rows of a 64-row tile of the right-hand side <- sources
struct PanelGroup { int col, local; };                  // workgroup -> (block column, its local workgroup index)
struct LevelPlan {
  int n_pad = 0, levels = 0, cols_total = 0;
  std::vector<int> new_of_old;                          // unknown -> row of the padded system
  std::vector<int> col_off, cols;                       // levels + 1; block columns in level order
  std::vector<int> row_off, row_tiles;                  // per block column (cols_total + 1): the 64-row tiles of its panel
  std::vector<int> pwg_off; std::vector<PanelGroup> pwg;  // levels + 1; the panel launch of a level: a workgroup per 8 rows of a column's panel tiles ...
  std::vector<int> pwt_off; std::vector<PanelGroup> pwt;  // ... or per whole 64-row tile (wide levels)
  std::vector<int> upd_off; std::vector<Target> targets; std::vector<int> sources;
  std::vector<int> fwd_off; std::vector<RowTarget> ftargets; std::vector<int> fsources;
  long long tile_updates = 0;
  double update_fraction = 1.0;
  bool ordered = false;
  // the dense tail (k_nd_tail): the levels main_levels .. levels - 1 hold exactly the block columns tail_col0 .. cols_total - 1, one per level in ascending order,
  // tail_col0 on a 64-row tile — the top separator, factorised by one launch instead of level by level.  No tail: tail_col0 = cols_total, main_levels = levels.
  int tail_col0 = 0, main_levels = 0;
  FlowPlan flow;                                        // the same factorisation as tasks of one launch (plan_flow)
};

// The longest run of one-column levels at the end of a schedule whose columns are the last ones of the matrix in ascending order, cut to a 64-row tile boundary
// (per_tile = block columns per 64-row tile); fewer than min_cols columns: no tail.
inline void plan_tail(const std::vector<int>& col_off, const std::vector<int>& cols, int cols_total, int per_tile, int min_cols, int* tail_col0, int* main_levels) {
  const int L = (int)col_off.size() - 1;
  int l = L, c = cols_total;
  while (l > 0 && col_off[(size_t)l] - col_off[(size_t)l - 1] == 1 && cols[(size_t)col_off[(size_t)l - 1]] == c - 1) { --l; --c; }
  while (c < cols_total && c % per_tile != 0) { ++c; ++l; }
  if (cols_total - c < min_cols) { c = cols_total; l = L; }
  *tail_col0 = c; *main_levels = l;
}

inline void plan_levels(int n, int n_blocks, const int* row_idx, const int* col_idx, int NB, int leaf_nodes, LevelPlan* P, int flow_chunk = 12) {
#ifdef PVLM_PLAN_PROFILE
  auto T0 = std::chrono::steady_clock::now(); auto mark = [&](const char* w) { auto t = std::chrono::steady_clock::now(); printf("%-28s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; };
#else
  auto mark = [](const char*) {};
#endif
  *P = LevelPlan();
  P->new_of_old.assign((size_t)std::max(n, 0), 0);
  if (n <= 0) return;
  // ---- nodes and their graph: as in plan_symbolic
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
  mark("nodes + graph");
  // ---- nested dissection: groups in elimination order.  `tag[v]` = id of the subset v currently belongs to (a subset is a list of nodes + its tag)
  std::vector<std::vector<int>> groups;
  {
    std::vector<int> tag((size_t)N, 0), lev((size_t)N, -1), queue;
    int next_tag = 1;
    // breadth-first levels of `start` inside the subset `t`; returns the visit order in `queue`
    auto bfs = [&](int start, int t) {
      queue.clear(); queue.push_back(start); lev[(size_t)start] = 0;
      for (size_t h = 0; h < queue.size(); ++h) { const int u = queue[h]; for (int w : adj[(size_t)u]) if (tag[(size_t)w] == t && lev[(size_t)w] < 0) { lev[(size_t)w] = lev[(size_t)u] + 1; queue.push_back(w); } }
    };
    struct Work { std::vector<int> nodes; int t; bool emit_after; };     // emit_after: a separator, emitted as a group when it comes off the stack
    std::vector<Work> stack;
    { Work w; w.nodes.resize((size_t)N); for (int v = 0; v < N; ++v) w.nodes[(size_t)v] = v; w.t = 0; w.emit_after = false; stack.push_back(std::move(w)); }
    while (!stack.empty()) {
      Work w = std::move(stack.back()); stack.pop_back();
      if (w.emit_after) { groups.push_back(std::move(w.nodes)); continue; }
      if (w.nodes.empty()) continue;
      if ((int)w.nodes.size() <= std::max(leaf_nodes, 1)) { groups.push_back(std::move(w.nodes)); continue; }
      const int t = w.t;
      for (int v : w.nodes) lev[(size_t)v] = -1;
      // connected components of the subset: several -> two bins, no separator
      bfs(w.nodes[0], t);
      if (queue.size() < w.nodes.size()) {
        std::vector<std::vector<int>> comps; comps.push_back(queue);
        for (int v : w.nodes) if (lev[(size_t)v] < 0) { bfs(v, t); comps.push_back(queue); }
        std::sort(comps.begin(), comps.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() != b.size() ? a.size() > b.size() : a[0] < b[0]; });
        Work A, B; A.t = next_tag++; B.t = next_tag++; A.emit_after = B.emit_after = false;
        for (auto& c : comps) { Work& dst = A.nodes.size() <= B.nodes.size() ? A : B; for (int v : c) { tag[(size_t)v] = dst.t; dst.nodes.push_back(v); } }
        stack.push_back(std::move(B)); stack.push_back(std::move(A));      // A is dissected (and emitted) first
        continue;
      }
      // pseudo-peripheral start: the farthest node of the farthest node ... (three rounds)
      int start = w.nodes[0];
      for (int round = 0; round < 3; ++round) {
        for (int v : w.nodes) lev[(size_t)v] = -1;
        bfs(start, t);
        int far = start;
        for (int v : queue) if (lev[(size_t)v] > lev[(size_t)far] || (lev[(size_t)v] == lev[(size_t)far] && adj[(size_t)v].size() < adj[(size_t)far].size())) far = v;
        if (far == start) break;
        start = far;
      }
      for (int v : w.nodes) lev[(size_t)v] = -1;
      bfs(start, t);
      int depth = 0;
      for (int v : w.nodes) depth = std::max(depth, lev[(size_t)v]);
      if (depth < 2) { groups.push_back(std::move(w.nodes)); continue; }   // (nearly) a clique: one dense group
      std::vector<int> count((size_t)depth + 1, 0);
      for (int v : w.nodes) ++count[(size_t)lev[(size_t)v]];
      long long best_cost = -1; int best = 1; int below = count[0];
      for (int l = 1; l < depth; ++l) {
        const int above = (int)w.nodes.size() - below - count[(size_t)l];
        const long long cost = std::llabs((long long)below - above) + 2ll * count[(size_t)l];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = l; }
        below += count[(size_t)l];
      }
      Work A, B, S; A.t = next_tag++; B.t = next_tag++; S.t = -1; A.emit_after = B.emit_after = false; S.emit_after = true;
      for (int v : w.nodes) if (lev[(size_t)v] > best) { tag[(size_t)v] = B.t; B.nodes.push_back(v); }
      for (int v : w.nodes) {
        if (lev[(size_t)v] < best) { tag[(size_t)v] = A.t; A.nodes.push_back(v); }
        else if (lev[(size_t)v] == best) {
          bool touches_b = false;                                           // a separator node without a neighbour on the far side belongs to the near side
          for (int q : adj[(size_t)v]) if (tag[(size_t)q] == B.t) { touches_b = true; break; }
          if (touches_b) { tag[(size_t)v] = -1; S.nodes.push_back(v); } else { tag[(size_t)v] = A.t; A.nodes.push_back(v); }
        }
      }
      stack.push_back(std::move(S)); stack.push_back(std::move(B)); stack.push_back(std::move(A));
    }
  }
  mark("nested dissection");
  // ---- rows of the padded system: every group starts on a 64-row tile
  int n_pad = 0;
  for (auto& g : groups) {
    std::sort(g.begin(), g.end());
    for (int v : g) for (int i : members[(size_t)v]) P->new_of_old[(size_t)i] = n_pad++;
    n_pad = (n_pad + 63) / 64 * 64;
  }
  P->n_pad = n_pad;
  // ---- symbolic factorisation on (64-row tile) x (NB-column block column) cells of the padded system, with the level of every block column
  const int C = n_pad / NB, T = n_pad / 64, per_tile = 64 / NB;
  P->cols_total = C;
  std::vector<unsigned char> nz((size_t)T * C, 0);
  for (int c = 0; c < C; ++c) nz[(size_t)(c / per_tile) * C + c] = 1;           // the diagonal (dummy rows are identity rows)
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
  mark("structure of the input");
  std::vector<int> level((size_t)C, 0);
  P->row_off.assign(1, 0);
  std::vector<int> R;
  long long dense_pairs = 0;
  for (int k = 0; k < C; ++k) {
    const int base = (k + 1) * NB, t0 = base / 64;
    R.clear();
    for (int t = t0; t < T && base < n_pad; ++t) if (nz[(size_t)t * C + k] && (t + 1) * 64 > base) R.push_back(t);
    for (int t : R) P->row_tiles.push_back(t);
    P->row_off.push_back((int)P->row_tiles.size());
    for (size_t a = 0; a < R.size(); ++a)
      for (size_t c = 0; c <= a; ++c)
        for (int col = per_tile * R[c]; col < per_tile * (R[c] + 1); ++col) if (col > k && col < C) nz[(size_t)R[a] * C + col] = 1;
    for (int t : R) for (int col = per_tile * t; col < per_tile * (t + 1); ++col) if (col > k && col < C) level[(size_t)col] = std::max(level[(size_t)col], level[(size_t)k] + 1);
    P->tile_updates += (long long)R.size() * ((long long)R.size() + 1) / 2;
    const long long dt = (n_pad - base + 63) / 64;
    dense_pairs += dt * (dt + 1) / 2;
  }
  P->update_fraction = dense_pairs ? (double)P->tile_updates / (double)dense_pairs : 1.0;
  mark("symbolic factorisation");
  // ---- the schedule
  int L = 0;
  for (int k = 0; k < C; ++k) L = std::max(L, level[(size_t)k] + 1);
  P->levels = L;
  P->col_off.assign((size_t)L + 1, 0);
  for (int k = 0; k < C; ++k) ++P->col_off[(size_t)level[(size_t)k] + 1];
  for (int l = 0; l < L; ++l) P->col_off[(size_t)l + 1] += P->col_off[(size_t)l];
  P->cols.assign((size_t)C, 0);
  { std::vector<int> cur(P->col_off.begin(), P->col_off.end() - 1); for (int k = 0; k < C; ++k) P->cols[(size_t)cur[(size_t)level[(size_t)k]]++] = k; }
  P->pwg_off.assign(1, 0); P->pwt_off.assign(1, 0); P->upd_off.assign(1, 0); P->fwd_off.assign(1, 0);
  // targets and their sources per level without sorting (160 000 tile updates on the Floor graph): a dense (tile, tile) -> target table, two passes over the level's
  // columns in ascending order — count, then fill — so that the sources of a target come out in ascending column order
  std::vector<int> slot((size_t)T * T, -1), rslot((size_t)T, -1), fill;
  P->pwg.reserve((size_t)P->row_tiles.size() * 8 + (size_t)C); P->pwt.reserve(P->row_tiles.size() + (size_t)C);
  P->sources.reserve((size_t)P->tile_updates); P->fsources.reserve(P->row_tiles.size());
  for (int l = 0; l < L; ++l) {
    const size_t t_first = P->targets.size(), f_first = P->ftargets.size();
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        int off = (int)P->sources.size();
        for (size_t q = t_first; q < P->targets.size(); ++q) { P->targets[q].src_off = off; off += P->targets[q].n_src; P->targets[q].n_src = 0; }
        P->sources.resize((size_t)off);
        off = (int)P->fsources.size();
        for (size_t q = f_first; q < P->ftargets.size(); ++q) { P->ftargets[q].src_off = off; off += P->ftargets[q].n_src; P->ftargets[q].n_src = 0; }
        P->fsources.resize((size_t)off);
      }
      for (int q = P->col_off[(size_t)l]; q < P->col_off[(size_t)l + 1]; ++q) {
        const int k = P->cols[(size_t)q];
        const int* rt = P->row_tiles.data() + P->row_off[(size_t)k]; const int nrt = P->row_off[(size_t)k + 1] - P->row_off[(size_t)k];
        if (pass == 0) {
          for (int g = 0; g < std::max(1, nrt * 8); ++g) P->pwg.push_back(PanelGroup{k, g});
          for (int g = 0; g < std::max(1, nrt); ++g) P->pwt.push_back(PanelGroup{k, g});
        }
        for (int a = 0; a < nrt; ++a) {
          int& rs = rslot[(size_t)rt[a]];
          if (pass == 0) { if (rs < 0) { rs = (int)P->ftargets.size(); P->ftargets.push_back(RowTarget{rt[a], 0, 0, 0}); } ++P->ftargets[(size_t)rs].n_src; }
          else { RowTarget& f = P->ftargets[(size_t)rs]; P->fsources[(size_t)(f.src_off + f.n_src++)] = k; }
          for (int c = 0; c <= a; ++c) {
            int& ts = slot[(size_t)rt[a] * T + rt[c]];
            if (pass == 0) {
              if (ts < 0) { ts = (int)P->targets.size(); P->targets.push_back(Target{rt[a], rt[c], 0, 0, 0, 0}); }
              Target& tg = P->targets[(size_t)ts];
              ++tg.n_src;
              if (k / per_tile == rt[c]) tg.col_min = std::max(tg.col_min, (k + 1) * NB);
            } else { Target& tg = P->targets[(size_t)ts]; P->sources[(size_t)(tg.src_off + tg.n_src++)] = k; }
          }
        }
      }
    }
    for (size_t q = t_first; q < P->targets.size(); ++q) slot[(size_t)P->targets[q].ti * T + P->targets[q].tj] = -1;
    for (size_t q = f_first; q < P->ftargets.size(); ++q) rslot[(size_t)P->ftargets[q].tile] = -1;
    P->pwg_off.push_back((int)P->pwg.size()); P->pwt_off.push_back((int)P->pwt.size()); P->upd_off.push_back((int)P->targets.size()); P->fwd_off.push_back((int)P->ftargets.size());
  }
  mark("schedule lists");
  plan_tail(P->col_off, P->cols, C, per_tile, 8, &P->tail_col0, &P->main_levels);
  plan_flow(n_pad, NB, P->row_off, P->row_tiles, flow_chunk, &P->flow);
  mark("flow tasks");
  P->ordered = true;
}

}  // namespace pvlm_spd
