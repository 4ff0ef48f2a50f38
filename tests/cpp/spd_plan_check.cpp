// Host-compiled check of csrc/pvlm_spd_plan.h (ordering + symbolic factorisation of the tile-sparse pose solve): returns the permutation and
// the per-block-column tile lists so that tests/test_spd_plan_cpu.py can hold them against a numeric Cholesky factor of the permuted matrix.
// TEST INFRASTRUCTURE ONLY.  Build: g++ -O2 -std=c++17 -fPIC -shared
#include <cstring>
#include "../../panovlm_amd/csrc/pvlm_spd_plan.h"

extern "C" {

// sizes[0] = row tiles, sizes[1] = pairs, sizes[2] = block columns; call with null outputs first to size them.
int chk_spd_plan(int n, int n_blocks, const int* row_idx, const int* col_idx, int nb, int* new_of_old, int* row_off, int* row_tiles, int* pair_off, int* pairs,
                 long long* sizes, double* update_fraction) {
  pvlm_spd::Symbolic S;
  pvlm_spd::plan_symbolic(n, n_blocks, row_idx, col_idx, nb, &S);
  sizes[0] = (long long)S.row_tiles.size(); sizes[1] = (long long)S.pairs.size(); sizes[2] = (long long)S.row_off.size() - 1;
  *update_fraction = S.update_fraction;
  if (new_of_old) std::memcpy(new_of_old, S.new_of_old.data(), (size_t)n * sizeof(int));
  if (row_off) std::memcpy(row_off, S.row_off.data(), S.row_off.size() * sizeof(int));
  if (pair_off) std::memcpy(pair_off, S.pair_off.data(), S.pair_off.size() * sizeof(int));
  if (row_tiles && !S.row_tiles.empty()) std::memcpy(row_tiles, S.row_tiles.data(), S.row_tiles.size() * sizeof(int));
  if (pairs && !S.pairs.empty()) std::memcpy(pairs, S.pairs.data(), S.pairs.size() * sizeof(pvlm_spd::TilePair));
  return S.ordered ? 1 : 0;
}

// the level plan (nested dissection + schedule, pvlm_spd::plan_levels).  First call with null outputs: sizes[0..9] = n_pad, levels, block columns, row tiles, panel
// groups, targets, sources, row targets, row sources, tile updates.  Second call fills the arrays (targets: 6 ints each, row targets: 4, panel groups: 2).
int chk_spd_levels(int n, int n_blocks, const int* row_idx, const int* col_idx, int nb, int leaf, long long* sizes, double* update_fraction, int* new_of_old, int* col_off,
                   int* cols, int* row_off, int* row_tiles, int* pwg_off, int* pwg, int* upd_off, int* targets, int* sources, int* fwd_off, int* ftargets, int* fsources) {
  pvlm_spd::LevelPlan P;
  pvlm_spd::plan_levels(n, n_blocks, row_idx, col_idx, nb, leaf, &P);
  const long long sz[10] = {P.n_pad, P.levels, P.cols_total, (long long)P.row_tiles.size(), (long long)P.pwg.size(), (long long)P.targets.size(), (long long)P.sources.size(),
                            (long long)P.ftargets.size(), (long long)P.fsources.size(), P.tile_updates};
  for (int k = 0; k < 10; ++k) sizes[k] = sz[k];
  *update_fraction = P.update_fraction;
  auto put = [](int* dst, const void* src, size_t bytes) { if (dst && bytes) std::memcpy(dst, src, bytes); };
  put(new_of_old, P.new_of_old.data(), P.new_of_old.size() * 4); put(col_off, P.col_off.data(), P.col_off.size() * 4); put(cols, P.cols.data(), P.cols.size() * 4);
  put(row_off, P.row_off.data(), P.row_off.size() * 4); put(row_tiles, P.row_tiles.data(), P.row_tiles.size() * 4);
  put(pwg_off, P.pwg_off.data(), P.pwg_off.size() * 4); put(pwg, P.pwg.data(), P.pwg.size() * sizeof(pvlm_spd::PanelGroup));
  put(upd_off, P.upd_off.data(), P.upd_off.size() * 4); put(targets, P.targets.data(), P.targets.size() * sizeof(pvlm_spd::Target)); put(sources, P.sources.data(), P.sources.size() * 4);
  put(fwd_off, P.fwd_off.data(), P.fwd_off.size() * 4); put(ftargets, P.ftargets.data(), P.ftargets.size() * sizeof(pvlm_spd::RowTarget)); put(fsources, P.fsources.data(), P.fsources.size() * 4);
  return P.ordered ? 1 : 0;
}

// the dense tail of a level plan (pvlm_spd::plan_tail with the given minimum number of columns; plan_levels itself asks for 8): out[0] = tail_col0, out[1] = main_levels,
// out[2] / out[3] = what plan_levels stored
void chk_spd_tail(int n, int n_blocks, const int* row_idx, const int* col_idx, int nb, int leaf, int min_cols, int* out) {
  pvlm_spd::LevelPlan P;
  pvlm_spd::plan_levels(n, n_blocks, row_idx, col_idx, nb, leaf, &P);
  pvlm_spd::plan_tail(P.col_off, P.cols, P.cols_total, 64 / nb, min_cols, &out[0], &out[1]);
  out[2] = P.tail_col0; out[3] = P.main_levels;
}

// the one-launch form of the level plan (pvlm_spd::plan_flow).  First call with null outputs: sizes[0..4] = tile columns, tasks, sources, below entries, depth; second call
// fills tasks (8 ints each: I, J, src_off, n_src, prev, final, 0, 0), sources (4 ints each), col_order, below_off, below; sizes[5] = finals.  Returns 1 when the flow plan is ready.
int chk_spd_flow(int n, int n_blocks, const int* row_idx, const int* col_idx, int nb, int leaf, long long* sizes, int* tasks, int* sources, int* col_order, int* below_off, int* below) {
  pvlm_spd::LevelPlan P;
  pvlm_spd::plan_levels(n, n_blocks, row_idx, col_idx, nb, leaf, &P);
  const pvlm_spd::FlowPlan& F = P.flow;
  sizes[0] = F.tile_cols; sizes[1] = (long long)F.tasks.size(); sizes[2] = (long long)F.sources.size(); sizes[3] = (long long)F.below.size(); sizes[4] = F.depth; sizes[5] = F.finals;
  auto put = [](int* dst, const void* src, size_t bytes) { if (dst && bytes) std::memcpy(dst, src, bytes); };
  put(tasks, F.tasks.data(), F.tasks.size() * sizeof(pvlm_spd::FlowTask)); put(sources, F.sources.data(), F.sources.size() * sizeof(pvlm_spd::FlowSource));
  put(col_order, F.col_order.data(), F.col_order.size() * 4); put(below_off, F.below_off.data(), F.below_off.size() * 4); put(below, F.below.data(), F.below.size() * 4);
  return F.ready ? 1 : 0;
}

}  // extern "C"
