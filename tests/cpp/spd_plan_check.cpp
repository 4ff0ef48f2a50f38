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

}  // extern "C"
