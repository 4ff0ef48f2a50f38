// pvlm_host_solver.cpp — part of the C++ host mirror (pvlm_host.hpp): the ceres:: surface the reference touches (Problem, loss functions, the LM driver that stands in for ceres::Solve), the functor factories of base/CostFunction.h, the Exchange factories.
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

// ================================================================================================
// ceres-like problem / solver
// ================================================================================================
namespace ceres_like {

static const int kStride[6] = {7, 7, 9, 9, 10, 12};

bool CostFunction::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
  Engine& e = Engine::Default();
  if (kind == kReprojKind) {   // {aa_cw, t_cw, point_3d}: a one-observation reprojection set
    const int64_t off[2] = {0, 1};
    const int cam = 0;
    pvlm_baset* bs = nullptr;
    if (pvlm_ba_create(e.ctx(), 1, 1, off, &cam, row.data(), parameters[2], weight, &bs) != PVLM_OK) return false;
    bool ok = pvlm_set_poses(e.ctx(), 1, parameters[0], parameters[1]) == PVLM_OK;
    double J[9];
    ok = ok && pvlm_ba_eval(e.ctx(), bs, residuals, jacobians ? J : nullptr) == PVLM_OK;
    pvlm_ba_destroy(e.ctx(), bs);
    if (ok && jacobians)
      for (int b = 0; b < 3; ++b)
        if (jacobians[b]) for (int k = 0; k < 3; ++k) jacobians[b][k] = J[3 * b + k];
    return ok && std::isfinite(residuals[0]);
  }
  double aa[6] = {parameters[0][0], parameters[0][1], parameters[0][2], parameters[2][0], parameters[2][1], parameters[2][2]};
  double t[6] = {parameters[1][0], parameters[1][1], parameters[1][2], parameters[3][0], parameters[3][1], parameters[3][2]};
  const int64_t off[2] = {0, 1};
  const int ref = 0, nei = 1;
  pvlm_resset* rs = nullptr;
  if (pvlm_resset_upload(e.ctx(), (pvlm_functor)kind, flags, weight, 1, 1, off, &ref, &nei, row.data(), kStride[kind], &rs) != PVLM_OK) return false;
  bool ok = pvlm_set_poses(e.ctx(), 2, aa, t) == PVLM_OK;
  double J[12];
  ok = ok && pvlm_eval(e.ctx(), rs, residuals, jacobians ? J : nullptr) == PVLM_OK;
  pvlm_resset_destroy(e.ctx(), rs);
  if (ok && jacobians)
    for (int b = 0; b < 4; ++b)
      if (jacobians[b]) for (int k = 0; k < 3; ++k) jacobians[b][k] = J[3 * b + k];
  return ok && std::isfinite(residuals[0]);
}

struct Problem::Impl {
  // parameter blocks: every double[3] the caller registered, in first-seen order
  std::unordered_map<double*, int> block_id;                 // only looked up, never iterated: ids are handed out in arrival order
  std::vector<double*> blocks;
  std::vector<bool> constant;
  // poses = (aa block, t block) pairs, in first-seen order
  std::unordered_map<unsigned long long, int> pose_id;       // key: block id of aa << 32 | block id of t
  std::vector<std::pair<int, int>> poses;
  // A group = one device residual set.  Host-built groups collect consecutive AddResidualBlock
  // calls with identical (kind, flags, weight, loss); consecutive blocks with the same pose pair
  // form a segment.  dev_* are the pose ids the device set uses (for sets handed in by the
  // association kernels these are the caller's list indices), ref/nei the Problem pose ids.
  struct Group {
    int kind = 0; unsigned flags = 0; double weight = 1.0; LossFunction* loss = nullptr;
    std::vector<double> rows; std::vector<int64_t> off; std::vector<int> ref, nei, dev_ref, dev_nei;
    pvlm_resset* set = nullptr; bool external = false;
    pvlm_neq* neq = nullptr; int dev_poses = 0;
    std::vector<int> ui, uj;            // unordered dev-id pairs of the neq structure
    std::vector<int> dev_to_pose;       // dev id -> Problem pose id (-1 unused)
  };
  std::vector<Group> groups;
  // Reprojection blocks (camera pose + free 3-D point), one group per (weight, loss).  At Solve the observations
  // are sorted by point and handed to the GPU (pvlm_baset); the point blocks are eliminated there.
  struct Bundle {
    double weight = 1.0; LossFunction* loss = nullptr;
    std::vector<int> obs_pose, obs_point;      // Problem pose id / parameter-block id of the point, insertion order
    std::vector<double> obs_bearing;           // 3 per observation (as handed to Create, un-normalised)
    pvlm_baset* set = nullptr;
    std::vector<int> point_blocks;             // device point index -> parameter-block id
    std::vector<int> dev_to_pose;              // device camera id -> Problem pose id
    std::vector<int> ui, uj;                   // co-visible device camera pairs (packed layout)
  };
  std::vector<Bundle> bundles;
  std::vector<bool> is_point;                  // per parameter block
  std::vector<CostFunction*> owned_costs;
  std::set<LossFunction*> owned_losses;
  int num_blocks = 0;

  int Block(double* p) {
    auto it = block_id.find(p);
    if (it != block_id.end()) return it->second;
    const int id = (int)blocks.size();
    block_id[p] = id; blocks.push_back(p); constant.push_back(false); is_point.push_back(false);
    return id;
  }
  int Pose(double* aa, double* t) {
    const std::pair<int, int> k(Block(aa), Block(t));
    const unsigned long long key = ((unsigned long long)(unsigned)k.first << 32) | (unsigned)k.second;
    auto it = pose_id.find(key);
    if (it != pose_id.end()) return it->second;
    const int id = (int)poses.size();
    pose_id[key] = id; poses.push_back(k);
    return id;
  }
};

Problem::Problem() : impl_(new Impl()) {}
Problem::~Problem() {
  Engine& e = Engine::Default();
  for (auto& g : impl_->groups) { if (g.neq) pvlm_neq_destroy(e.ctx(), g.neq); if (g.set) pvlm_resset_destroy(e.ctx(), g.set); }
  for (auto& b : impl_->bundles) if (b.set) pvlm_ba_destroy(e.ctx(), b.set);
  for (CostFunction* c : impl_->owned_costs) delete c;
  for (LossFunction* l : impl_->owned_losses) delete l;
  delete impl_;
}
int Problem::NumResidualBlocks() const { return impl_->num_blocks; }
void Problem::RegisterPoses(std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list) {
  for (size_t i = 0; i < aa_list.size() && i < t_list.size(); ++i) impl_->Pose(aa_list[i].data(), t_list[i].data());
}
void Problem::SetParameterBlockConstant(double* block) { impl_->constant[impl_->Block(block)] = true; }

void Problem::AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n) {
  Impl& I = *impl_;
  const int pr = I.Pose(aa_r, t_r), pn = I.Pose(aa_n, t_n);
  if (loss) I.owned_losses.insert(loss);
  I.owned_costs.push_back(cost);
  // blocks are grouped by (functor, flags, weight, loss) regardless of the order they arrive in
  // (AddCameraLidarResidual alternates two functors); inside a group the insertion order is kept.
  int gi = -1;
  for (int k = (int)I.groups.size() - 1; k >= 0; --k) {
    const Impl::Group& c = I.groups[k];
    if (!c.external && !c.set && c.kind == cost->kind && c.flags == cost->flags && c.weight == cost->weight && c.loss == loss) { gi = k; break; }
  }
  if (gi < 0) {
    Impl::Group g; g.kind = cost->kind; g.flags = cost->flags; g.weight = cost->weight; g.loss = loss; g.off.push_back(0);
    I.groups.push_back(g);
    gi = (int)I.groups.size() - 1;
  }
  Impl::Group& g = I.groups[gi];
  if (g.ref.empty() || g.ref.back() != pr || g.nei.back() != pn) { g.ref.push_back(pr); g.nei.push_back(pn); g.off.push_back(g.off.back()); }
  g.rows.insert(g.rows.end(), cost->row.begin(), cost->row.end());
  g.off.back() += 1;
  I.num_blocks++;
}

void Problem::AddResidualRows(int kind, unsigned flags, double weight, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n,
                              const double* rows, size_t n) {
  if (n == 0) return;
  Impl& I = *impl_;
  const int pr = I.Pose(aa_r, t_r), pn = I.Pose(aa_n, t_n);
  if (loss) I.owned_losses.insert(loss);
  int gi = -1;
  for (int k = (int)I.groups.size() - 1; k >= 0; --k) {
    const Impl::Group& c = I.groups[k];
    if (!c.external && !c.set && c.kind == kind && c.flags == flags && c.weight == weight && c.loss == loss) { gi = k; break; }
  }
  if (gi < 0) {
    Impl::Group g; g.kind = kind; g.flags = flags; g.weight = weight; g.loss = loss; g.off.push_back(0);
    I.groups.push_back(g);
    gi = (int)I.groups.size() - 1;
  }
  Impl::Group& g = I.groups[gi];
  if (g.ref.empty() || g.ref.back() != pr || g.nei.back() != pn) { g.ref.push_back(pr); g.nei.push_back(pn); g.off.push_back(g.off.back()); }
  g.rows.insert(g.rows.end(), rows, rows + n * (size_t)kStride[kind]);
  g.off.back() += (int64_t)n;
  I.num_blocks += (int)n;
}

void Problem::AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_c, double* t_c, double* point_3d) {
  Impl& I = *impl_;
  if (cost->kind != kReprojKind) throw std::runtime_error("three-block AddResidualBlock expects PanoramaReprojResidual_1Angle");
  const int pose = I.Pose(aa_c, t_c);
  const int pb = I.Block(point_3d);
  I.is_point[pb] = true;
  if (loss) I.owned_losses.insert(loss);
  I.owned_costs.push_back(cost);
  int bi = -1;
  for (int k = (int)I.bundles.size() - 1; k >= 0; --k)
    if (!I.bundles[k].set && I.bundles[k].weight == cost->weight && I.bundles[k].loss == loss) { bi = k; break; }
  if (bi < 0) { Impl::Bundle b; b.weight = cost->weight; b.loss = loss; I.bundles.push_back(b); bi = (int)I.bundles.size() - 1; }
  Impl::Bundle& b = I.bundles[bi];
  b.obs_pose.push_back(pose); b.obs_point.push_back(pb);
  b.obs_bearing.insert(b.obs_bearing.end(), cost->row.begin(), cost->row.begin() + 3);
  I.num_blocks++;
}

void Problem::AddResidualSet(pvlm_resset* set, LossFunction* loss, std::vector<Vector3d>* aa_list, std::vector<Vector3d>* t_list) {
  Impl& I = *impl_;
  if (loss) I.owned_losses.insert(loss);
  int64_t n = 0; int P = 0, kind = 0; unsigned flags = 0;
  pvlm_resset_info(set, &n, &P, &kind, &flags);
  Impl::Group g; g.kind = kind; g.flags = flags; g.loss = loss; g.set = set; g.external = true;
  g.off.resize((size_t)P + 1); g.dev_ref.resize((size_t)std::max(P, 1)); g.dev_nei.resize((size_t)std::max(P, 1));
  Engine& e = Engine::Default();
  e.Check(pvlm_resset_download(e.ctx(), set, g.off.data(), g.dev_ref.data(), g.dev_nei.data(), nullptr), "pvlm_resset_download");
  g.dev_ref.resize(P); g.dev_nei.resize(P);
  for (int p = 0; p < P; ++p) {
    g.ref.push_back(I.Pose((*aa_list)[g.dev_ref[p]].data(), (*t_list)[g.dev_ref[p]].data()));
    g.nei.push_back(I.Pose((*aa_list)[g.dev_nei[p]].data(), (*t_list)[g.dev_nei[p]].data()));
  }
  I.groups.push_back(g);
  I.num_blocks += (int)n;
}

std::string Solver::Summary::BriefReport() const {
  char b[256];
  snprintf(b, sizeof(b), "pvlm LM: blocks %d, initial cost %.6e, final cost %.6e, successful %d, unsuccessful %d, %s", num_residual_blocks,
           initial_cost, final_cost, num_successful_steps, num_unsuccessful_steps, message.c_str());
  return b;
}

namespace {

// Skyline (profile) Cholesky of a symmetric positive definite matrix given as dense row-major
// lower triangle accessor.  Pose graphs of LiDAR odometry are block-banded (temporal neighbours)
// plus a few loop closures, which the envelope captures.
struct Skyline {
  int n = 0;
  std::vector<int> first;          // first stored column of each row
  std::vector<size_t> start;       // offset of row i in val (entries first[i]..i)
  std::vector<double> val;
  double& at(int i, int j) { return val[start[i] + (size_t)(j - first[i])]; }
  void Init(const std::vector<int>& f) {
    n = (int)f.size(); first = f; start.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) start[i + 1] = start[i] + (size_t)(i - first[i] + 1);
    val.assign(start[n], 0.0);
  }
  bool Factor() {
    for (int i = 0; i < n; ++i) {
      for (int j = first[i]; j <= i; ++j) {
        double s = at(i, j);
        const int k0 = std::max(first[i], first[j]);
        for (int k = k0; k < j; ++k) s -= at(i, k) * at(j, k);
        if (j < i) at(i, j) = s / at(j, j);
        else { if (!(s > 0.0)) return false; at(i, i) = std::sqrt(s); }
      }
    }
    return true;
  }
  void Solve(std::vector<double>& b) {
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = first[i]; k < i; ++k) s -= at(i, k) * b[k]; b[i] = s / at(i, i); }
    for (int i = n - 1; i >= 0; --i) { b[i] /= at(i, i); for (int k = first[i]; k < i; ++k) b[k] -= at(i, k) * b[i]; }
  }
};

// Gauss-Newton blocks of the four-block groups at one point.  The block STRUCTURE is fixed for a whole Solve (the sorted key
// list is built once and shared); an evaluation only refills the numbers: H[36 k ...] = block keys[k] = (pose a <= pose b),
// 6x6 row-major d2/dx_a dx_b.  (Round 2 rebuilt a std::map of 36-double nodes at every evaluation: 1.6 ms per LM step at Room scale.)
using BlockKeys = std::vector<std::pair<int, int>>;
struct Assembled {
  double cost = 0;
  std::vector<double> g;                       // n_free
  std::shared_ptr<const BlockKeys> keys;
  std::vector<double> H;                       // 36 per key
};
// one iteration protocol for the flat table and for the std::map the reprojection path still uses
template <typename F> inline void ForEachBlock(const Assembled& A, F&& f) {
  if (!A.keys) return;
  for (size_t k = 0; k < A.keys->size(); ++k) f((*A.keys)[k], A.H.data() + 36 * k);
}
template <typename F> inline void ForEachBlock(const std::map<std::pair<int, int>, std::array<double, 36>>& H, F&& f) {
  for (auto& kv : H) f(kv.first, kv.second.data());
}

}  // namespace

void Solve(const Solver::Options& opt, Problem* problem, Solver::Summary* summary) {
  StageTimer stage_timer_("solve (LM)");
  Problem::Impl& I = *problem->impl();
  Engine& e = Engine::Default();
  *summary = Solver::Summary();
  summary->num_residual_blocks = I.num_blocks;
  const int NP = (int)I.poses.size();
  const Exchange* xch = (opt.exchange && opt.exchange->active()) ? opt.exchange : nullptr;
  // a rank of a sharded solve may own no residual block at all and still has to take part in every exchange
  if (!xch && (I.num_blocks == 0 || NP == 0)) { summary->message = "no residual blocks"; return; }
  if (xch) {
    // Entry into a sharded Solve is collective: a rank that cannot take part (no registered poses, reprojection blocks — which are
    // not sharded) must not leave its peers waiting in the first all-reduce.  Every rank contributes its own verdict to one
    // exchange and all of them leave together, with the same message.
    bool has_bundle = false;
    for (auto& b : I.bundles) if (!b.obs_pose.empty()) has_bundle = true;
    // sum and sum of squares of the pose counts: world * sum(NP^2) == sum(NP)^2 holds exactly when all counts are equal (Cauchy-Schwarz),
    // and every rank evaluates the same reduced numbers — a test against the local NP alone let the rank whose count equals the mean
    // pass while its peers threw, and hang in the next exchange
    double agreed[4] = {NP == 0 ? 1.0 : 0.0, has_bundle ? 1.0 : 0.0, (double)NP, (double)NP * (double)NP};
    xch->allreduce_sum(agreed, 4);
    if (agreed[0] > 0) throw std::runtime_error("sharded Solve: " + std::to_string((int)agreed[0]) + " rank(s) entered without registered poses (Problem::RegisterPoses must run on every rank)");
    if (agreed[1] > 0) throw std::runtime_error("sharded Solve: reprojection blocks are not sharded (camera terms run on one GPU)");
    if ((double)xch->world * agreed[3] != agreed[2] * agreed[2]) throw std::runtime_error("sharded Solve: the ranks registered different numbers of poses");
  }
  // concatenation of every rank's list through the one primitive an Exchange has
  auto all_concat = [&](const std::vector<double>& mine) {
    std::vector<double> cnt((size_t)xch->world, 0.0);
    cnt[(size_t)xch->rank] = (double)mine.size();
    xch->allreduce_sum(cnt.data(), cnt.size());
    size_t total = 0, off = 0;
    for (int r = 0; r < xch->world; ++r) { if (r == xch->rank) off = total; total += (size_t)cnt[(size_t)r]; }
    std::vector<double> all(std::max<size_t>(total, 1), 0.0);
    std::copy(mine.begin(), mine.end(), all.begin() + (std::ptrdiff_t)off);
    xch->allreduce_sum(all.data(), all.size());
    all.resize(total);
    return all;
  };

  // ---- device sets, ONE pose numbering and ONE normal-equation structure for all four-block groups --------------
  // Every group's segments are renumbered to the Problem's pose ids (a set that came from the association carries the caller's list
  // indices: pvlm_resset_set_pose_ids, once), so that a linearisation needs one pose table and the groups' blocks are summed on the
  // device into one packed buffer [diag NP x 36 | off U x 36 | g NP x 6 | cost] over the union (ui < uj) of their pose pairs.
  StageTimer* stage_timer_setup_ = new StageTimer("solve: residual-set upload + structures");
  const char* gpu_min_env = std::getenv("PVLM_GPU_CHOLESKY_MIN");
  const int gpu_chol_min = gpu_min_env ? std::atoi(gpu_min_env) : 1500;
  // The plan of the pose solve (ordering + schedule of pvlm_spd_solve_blocks: ~10 ms of host work at Floor size) is asked for as soon as the block structure is
  // known and made on a thread of the library beside what follows — set uploads, structures, the first linearisation — instead of inside the first Cholesky of the
  // Solve.  A hint: pvlm_spd_solve_blocks compares the lists it is given with the prefetched ones and plans for itself when they differ.
  auto prefetch_plan = [&](const BlockKeys& keys, const std::vector<int>& block_off, int n_free) {
    if (n_free < gpu_chol_min || std::getenv("PVLM_NO_PLAN_PREFETCH")) return;
    std::vector<int> rows, cols, mirror;
    rows.reserve(keys.size() * 6); cols.reserve(keys.size() * 6); mirror.reserve(keys.size());
    auto at = [&](int pose, int r) { const int b = r < 3 ? I.poses[(size_t)pose].first : I.poses[(size_t)pose].second; return block_off[(size_t)b] < 0 ? -1 : block_off[(size_t)b] + (r % 3); };
    for (const auto& key : keys) {
      for (int r = 0; r < 6; ++r) { rows.push_back(at(key.first, r)); cols.push_back(at(key.second, r)); }
      mirror.push_back(key.first != key.second ? 1 : 0);
    }
    e.Check(pvlm_spd_plan_prefetch(e.ctx(), n_free, (int)mirror.size(), rows.data(), cols.data(), mirror.data()), "pvlm_spd_plan_prefetch");
  };
  bool any_bundle = false;
  for (auto& b : I.bundles) if (!b.obs_pose.empty()) any_bundle = true;
  // the union (ui < uj) of the groups' pose pairs, sorted: made once (a sorted vector: 35 000 insertions into a std::set were 3 ms of this stage at Floor size)
  std::vector<std::pair<int, int>> up;
  {
    size_t total = 0;
    for (auto& g : I.groups) total += g.ref.size();
    up.reserve(total);
    for (auto& g : I.groups) for (size_t p = 0; p < g.ref.size(); ++p) up.push_back({std::min(g.ref[p], g.nei[p]), std::max(g.ref[p], g.nei[p])});
    std::sort(up.begin(), up.end());
    up.erase(std::unique(up.begin(), up.end()), up.end());
  }
  if (!xch && !any_bundle) {
    // one process, four-block groups only: the key list is [every pose with itself | the union of the groups' pose pairs, sorted] and every block is free unless
    // constant — the same lists the code below arrives at (it stays the authority)
    std::vector<int> off(I.blocks.size(), -1);
    int nf = 0;
    for (int p = 0; p < NP; ++p)
      for (int b : {I.poses[(size_t)p].first, I.poses[(size_t)p].second})
        if (!I.constant[(size_t)b] && off[(size_t)b] < 0) { off[(size_t)b] = nf; nf += 3; }
    BlockKeys early;
    early.reserve((size_t)NP + up.size());
    for (int p = 0; p < NP; ++p) early.push_back({p, p});
    for (auto& u : up) early.push_back(u);
    prefetch_plan(early, off, nf);
  }
  std::vector<int> gui, guj;
  {
    for (auto& g : I.groups) {
      const int P = (int)g.ref.size();
      if (!g.set) {
        g.dev_ref = g.ref; g.dev_nei = g.nei;
        e.Check(pvlm_resset_upload(e.ctx(), (pvlm_functor)g.kind, g.flags, g.weight, g.off.back(), P, g.off.data(), g.dev_ref.data(), g.dev_nei.data(),
                                   g.rows.data(), kStride[g.kind], &g.set), "pvlm_resset_upload");
        std::vector<double>().swap(g.rows);
      } else if (g.dev_ref != g.ref || g.dev_nei != g.nei) {
        e.Check(pvlm_resset_set_pose_ids(e.ctx(), g.set, g.ref.data(), g.nei.data()), "pvlm_resset_set_pose_ids");
        g.dev_ref = g.ref; g.dev_nei = g.nei;
      }
    }
    gui.reserve(up.size()); guj.reserve(up.size());
    for (auto& u : up) { gui.push_back(u.first); guj.push_back(u.second); }
    for (auto& g : I.groups) {
      if (g.neq && g.dev_poses == NP && g.ui == gui && g.uj == guj) continue;      // a second Solve on an unchanged Problem
      if (g.neq) { pvlm_neq_destroy(e.ctx(), g.neq); g.neq = nullptr; }
      g.dev_poses = NP; g.ui = gui; g.uj = guj;
      g.dev_to_pose.resize((size_t)NP);
      for (int p = 0; p < NP; ++p) g.dev_to_pose[(size_t)p] = p;
      e.Check(pvlm_neq_create(e.ctx(), NP, (int)gui.size(), gui.data(), guj.data(), &g.neq), "pvlm_neq_create");
    }
  }
  delete stage_timer_setup_;
  // ---- reprojection sets: observations sorted by point, points resident on the GPU ------------------
  bool have_bundles = false;
  for (auto& b : I.bundles) {
    if (b.obs_pose.empty()) continue;
    have_bundles = true;
    if (b.set) continue;
    StageTimer stage_timer_bundle_("solve: reprojection set creation (host grouping by point + pvlm_ba_create)");
    std::unordered_map<int, int> pidx, cidx;      // ids are handed out in order of first appearance: the container's order plays no role
    const size_t n = b.obs_pose.size();
    std::vector<int> obs_dev_point(n);
    for (size_t i = 0; i < n; ++i) {
      auto it = pidx.find(b.obs_point[i]);
      if (it == pidx.end()) { it = pidx.insert({b.obs_point[i], (int)b.point_blocks.size()}).first; b.point_blocks.push_back(b.obs_point[i]); }
      obs_dev_point[i] = it->second;
    }
    const int M = (int)b.point_blocks.size();
    std::vector<int64_t> off((size_t)M + 1, 0);
    for (size_t i = 0; i < n; ++i) off[(size_t)obs_dev_point[i] + 1]++;
    for (int p = 0; p < M; ++p) off[(size_t)p + 1] += off[p];
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    std::vector<int> cam(n); std::vector<double> bearing(3 * n), points((size_t)M * 3);
    for (size_t i = 0; i < n; ++i) {   // stable: insertion order inside a point's track
      const size_t dst = (size_t)fill[obs_dev_point[i]]++;
      auto ic = cidx.find(b.obs_pose[i]);
      if (ic == cidx.end()) { ic = cidx.insert({b.obs_pose[i], (int)b.dev_to_pose.size()}).first; b.dev_to_pose.push_back(b.obs_pose[i]); }
      cam[dst] = ic->second;
      for (int k = 0; k < 3; ++k) bearing[3 * dst + k] = b.obs_bearing[3 * i + k];
    }
    std::vector<unsigned char> frozen((size_t)M, 0); bool any_frozen = false;
    for (int p = 0; p < M; ++p) {
      for (int k = 0; k < 3; ++k) points[(size_t)p * 3 + k] = I.blocks[b.point_blocks[p]][k];
      if (I.constant[b.point_blocks[p]]) { frozen[p] = 1; any_frozen = true; }
    }
    e.Check(pvlm_ba_create(e.ctx(), M, (int64_t)n, off.data(), cam.data(), bearing.data(), points.data(), b.weight, &b.set), "pvlm_ba_create");
    if (any_frozen) e.Check(pvlm_ba_set_constant(e.ctx(), b.set, frozen.data()), "pvlm_ba_set_constant");
    int nu = 0;
    pvlm_ba_structure(b.set, nullptr, nullptr, nullptr, &nu, nullptr, nullptr);
    b.ui.resize(nu); b.uj.resize(nu);
    pvlm_ba_structure(b.set, nullptr, nullptr, nullptr, nullptr, b.ui.data(), b.uj.data());
    std::vector<int>().swap(b.obs_pose); std::vector<int>().swap(b.obs_point); std::vector<double>().swap(b.obs_bearing);
    b.obs_pose.push_back(-1);   // keeps "non-empty" for later Solve calls on the same Problem
  }

  // ---- free-parameter layout (pose blocks; the point blocks never reach the host system) -------------
  std::vector<int> block_off(I.blocks.size(), -1);
  int n_free = 0;
  // sharded solve: the poses were registered up front (RegisterPoses), so a rank also knows poses none of ITS blocks
  // touch; a pose no rank touches is left out of the system, as if it had never been added
  std::vector<char> touched(I.blocks.size(), xch ? 0 : 1);
  std::vector<std::pair<int, int>> xkeys;          // sharded solve: sorted union of the ranks' 6x6 block keys (pose a <= pose b)
  if (xch) {
    std::vector<double> use(I.blocks.size(), 0.0), keys;
    std::set<std::pair<int, int>> mine;
    for (auto& g : I.groups)
      for (size_t p = 0; p < g.ref.size(); ++p) {
        for (int q : {g.ref[p], g.nei[p]}) { use[(size_t)I.poses[q].first] = 1.0; use[(size_t)I.poses[q].second] = 1.0; mine.insert({q, q}); }
        mine.insert({std::min(g.ref[p], g.nei[p]), std::max(g.ref[p], g.nei[p])});
      }
    xch->allreduce_sum(use.data(), use.size());
    for (size_t b = 0; b < use.size(); ++b) touched[b] = use[b] > 0.0;
    for (auto& k : mine) { keys.push_back((double)k.first); keys.push_back((double)k.second); }
    const std::vector<double> all = all_concat(keys);
    std::set<std::pair<int, int>> uni;
    for (size_t i = 0; i + 1 < all.size(); i += 2) uni.insert({(int)all[i], (int)all[i + 1]});
    xkeys.assign(uni.begin(), uni.end());
  }
  for (int p = 0; p < NP; ++p)
    for (int b : {I.poses[p].first, I.poses[p].second})
      if (!I.constant[b] && touched[b] && block_off[b] < 0) { block_off[b] = n_free; n_free += 3; }
  if (n_free == 0) { summary->message = "all parameter blocks constant"; }

  std::vector<double> x(3 * I.blocks.size());
  auto load_x = [&]() { for (size_t b = 0; b < I.blocks.size(); ++b) for (int k = 0; k < 3; ++k) x[3 * b + k] = I.blocks[b][k]; };
  auto store_x = [&](const std::vector<double>& v) { for (size_t b = 0; b < I.blocks.size(); ++b) for (int k = 0; k < 3; ++k) I.blocks[b][k] = v[3 * b + k]; };
  load_x();
  // scalar row/col index of (pose, half, k)
  auto idx = [&](int pose, int r) { const int b = r < 3 ? I.poses[pose].first : I.poses[pose].second; return block_off[b] < 0 ? -1 : block_off[b] + (r % 3); };

  // ---- fixed block structure of the four-block groups (built once per Solve) ----------------------------------------
  // Unsharded: the key list IS the packed layout — the NP diagonal blocks, then the U pair blocks — so an evaluation's table is a plain
  // copy of the buffer the GPU filled.  Sharded: the sorted union over the ranks (the exchanged buffer is then the table), filled
  // through a slot map.
  const int n_groups = (int)I.groups.size();
  const int U = (int)gui.size();
  auto keys = std::make_shared<BlockKeys>();
  std::vector<int> slot_of_packed;              // sharded: packed block (diag p | pair u) -> position in the key list
  if (xch) {
    *keys = xkeys;
    auto slot_of = [&](int a, int b) {
      const auto it = std::lower_bound(keys->begin(), keys->end(), std::make_pair(a, b));
      if (it == keys->end() || *it != std::make_pair(a, b)) throw std::runtime_error("Solve: block key missing from the structure");
      return (int)(it - keys->begin());
    };
    std::vector<char> used((size_t)NP, 0);
    for (auto& g : I.groups) for (size_t p = 0; p < g.ref.size(); ++p) { used[(size_t)g.ref[p]] = 1; used[(size_t)g.nei[p]] = 1; }
    slot_of_packed.assign((size_t)NP + (size_t)U, -1);
    for (int p = 0; p < NP; ++p) if (used[(size_t)p]) slot_of_packed[(size_t)p] = slot_of(p, p);
    for (int u = 0; u < U; ++u) slot_of_packed[(size_t)NP + (size_t)u] = slot_of(gui[(size_t)u], guj[(size_t)u]);
    if (!have_bundles) prefetch_plan(*keys, block_off, n_free);        // sharded: the structure is known after the exchange of the key lists; the first linearisation is still ahead
  } else {
    keys->reserve((size_t)NP + (size_t)U);
    for (int p = 0; p < NP; ++p) keys->push_back({p, p});
    for (int u = 0; u < U; ++u) keys->push_back({gui[(size_t)u], guj[(size_t)u]});
  }
  // pinned landing buffer of the packed normal equations; released on every exit path (Solve has several)
  struct PackedIO {
    double* packed = nullptr; size_t count = 0; pvlm_ctx* ctx;
    std::vector<double> aa, tt;
    std::vector<pvlm_neq*> neq; std::vector<const pvlm_resset*> sets; std::vector<pvlm_loss> loss; std::vector<double> loss_a;
    explicit PackedIO(pvlm_ctx* c) : ctx(c) {}
    ~PackedIO() { pvlm_synchronize(ctx); if (packed) pvlm_host_free(ctx, packed); }
  } io(e.ctx());
  if (n_groups > 0) {
    io.count = (size_t)pvlm_neq_size(I.groups[0].neq);
    void* p = nullptr;
    e.Check(pvlm_host_alloc(e.ctx(), (int64_t)(io.count * sizeof(double)), &p), "pvlm_host_alloc");
    io.packed = static_cast<double*>(p);
    for (auto& g : I.groups) {
      io.neq.push_back(g.neq); io.sets.push_back(g.set);
      io.loss.push_back(g.loss ? (pvlm_loss)g.loss->kind() : PVLM_LOSS_NONE); io.loss_a.push_back(g.loss ? g.loss->a() : 0.0);
    }
  }
  io.aa.assign((size_t)NP * 3, 0.0); io.tt.assign((size_t)NP * 3, 0.0);

  // evaluates cost (+ H, g when want_H) of the four-block groups at parameter vector v: ONE pose table, ONE submission for all groups
  // (per group: pair table, fused kernel, epilogue, gather ADDING into the shared packed buffer), one queued copy, ONE synchronisation.
  // The groups' blocks are summed in group order, as the host used to add them.
  long evaluations = 0;
  auto evaluate = [&](const std::vector<double>& v, bool want_H, Assembled& A) {
    // the first linearisation of a Solve binds the structures to the residual sets (CSR upload), sizes the per-structure buffers
    // and, once per process, loads the kernels' code objects: timed apart from the steady LM steps
    StageTimer stage_timer_eval_(evaluations++ == 0 ? "solve: first linearisation of a Solve (structures bound, buffers sized, code objects loaded)"
                                                    : "solve: GPU linearisation + block assembly");
    static const bool eval_trace = std::getenv("PVLM_HOST_EVAL_TRACE") != nullptr;   // per-call phase times on stderr (profiling tools)
    const auto tr0 = std::chrono::steady_clock::now();
    A.cost = 0; A.g.assign(n_free, 0.0); A.keys = keys; A.H.clear();
    if (n_groups > 0) {
      for (int p = 0; p < NP; ++p)
        for (int k = 0; k < 3; ++k) { io.aa[3 * (size_t)p + k] = v[3 * I.poses[p].first + k]; io.tt[3 * (size_t)p + k] = v[3 * I.poses[p].second + k]; }
      e.Check(pvlm_set_poses(e.ctx(), NP, io.aa.data(), io.tt.data()), "pvlm_set_poses");
      e.Check(pvlm_neq_accumulate_sets(e.ctx(), n_groups, io.neq.data(), io.sets.data(), io.loss.data(), io.loss_a.data(), io.packed), "pvlm_neq_accumulate_sets");
    }
    const auto tr1 = std::chrono::steady_clock::now();
    if (n_groups > 0) e.Check(pvlm_synchronize(e.ctx()), "pvlm_synchronize");
    const auto tr2 = std::chrono::steady_clock::now();
    if (n_groups > 0) {
      const double* packed = io.packed;
      A.cost = packed[io.count - 1];
      if (want_H) {
        const double* gg = packed + ((size_t)NP + (size_t)U) * 36;
        for (int p = 0; p < NP; ++p)
          for (int half = 0; half < 2; ++half) {
            const int b = half ? I.poses[p].second : I.poses[p].first;
            if (block_off[b] >= 0) for (int k = 0; k < 3; ++k) A.g[block_off[b] + k] += gg[(size_t)p * 6 + 3 * half + k];
          }
        if (!xch) A.H.assign(packed, packed + ((size_t)NP + (size_t)U) * 36);
        else {
          A.H.assign(36 * keys->size(), 0.0);
          for (size_t q = 0; q < slot_of_packed.size(); ++q)
            if (slot_of_packed[q] >= 0) std::copy(packed + 36 * q, packed + 36 * (q + 1), A.H.begin() + 36 * (std::ptrdiff_t)slot_of_packed[q]);
        }
      }
    } else if (want_H) A.H.assign(36 * keys->size(), 0.0);
    if (eval_trace) {
      const auto tr3 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      fprintf(stderr, "[eval %ld] submission %.0f us, synchronisation %.0f us, table %.0f us (%d groups, %zu keys)\n", evaluations, us(tr0, tr1), us(tr1, tr2), us(tr2, tr3),
              n_groups, keys->size());
    }
    if (xch) {
      // the one exchange of an evaluation: [cost | g | blocks in key order], summed over the ranks;
      // afterwards every rank holds the same Assembled, bit for bit
      StageTimer stage_timer_x_("solve: exchange of the normal equations");
      std::vector<double> buf(want_H ? 1 + (size_t)n_free + A.H.size() : 1, 0.0);
      buf[0] = A.cost;
      if (want_H) { std::copy(A.g.begin(), A.g.end(), buf.begin() + 1); std::copy(A.H.begin(), A.H.end(), buf.begin() + 1 + n_free); }
      xch->allreduce_sum(buf.data(), buf.size());
      A.cost = buf[0];
      if (want_H) { std::copy(buf.begin() + 1, buf.begin() + 1 + n_free, A.g.begin()); std::copy(buf.begin() + 1 + n_free, buf.end(), A.H.begin()); }
    }
  };

  // ---- reprojection sets: reduced camera system for a given trust-region radius ------------------------
  struct Reduced {
    double cost = 0, gmax_points = 0;
    std::vector<double> g_red, g_cam, Udiag;                  // n_free each
    std::map<std::pair<int, int>, std::array<double, 36>> H;  // Schur complement blocks (pose a <= pose b)
  };
  auto bundle_poses = [&](const Problem::Impl::Bundle& b, const std::vector<double>& v) {
    const int nd = (int)b.dev_to_pose.size();
    std::vector<double> aa((size_t)nd * 3), tt((size_t)nd * 3);
    for (int d = 0; d < nd; ++d) {
      const int p = b.dev_to_pose[d];
      for (int k = 0; k < 3; ++k) { aa[3 * d + k] = v[3 * I.poses[p].first + k]; tt[3 * d + k] = v[3 * I.poses[p].second + k]; }
    }
    e.Check(pvlm_set_poses(e.ctx(), nd, aa.data(), tt.data()), "pvlm_set_poses");
  };
  auto bundle_reduce = [&](const std::vector<double>& v, double radius, bool init, Reduced& R) {
    StageTimer stage_timer_br_("solve: reprojection blocks reduced on the GPU + host scatter of the camera system");
    R = Reduced(); R.g_red.assign(n_free, 0.0); R.g_cam.assign(n_free, 0.0); R.Udiag.assign(n_free, 0.0);
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      bundle_poses(b, v);
      std::vector<double> packed((size_t)pvlm_ba_packed_size(b.set), 0.0);
      e.Check(pvlm_ba_reduce(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, init ? 1 : 0, radius,
                             opt.min_lm_diagonal, opt.max_lm_diagonal, packed.data()), "pvlm_ba_reduce");
      const int nd = (int)b.dev_to_pose.size(), nu = (int)b.ui.size();
      const double* Hd = packed.data(); const double* Ho = Hd + (size_t)nd * 36; const double* gg = Ho + (size_t)nu * 36;
      const double* Ud = gg + (size_t)nd * 6 + 1; const double* gc = Ud + (size_t)nd * 6;
      R.cost += gg[(size_t)nd * 6];
      R.gmax_points = std::max(R.gmax_points, packed.back());
      for (int d = 0; d < nd; ++d) {
        const int p = b.dev_to_pose[d];
        auto& blk = R.H[{p, p}];
        for (int k = 0; k < 36; ++k) blk[k] += Hd[(size_t)d * 36 + k];
        for (int r = 0; r < 6; ++r) {
          const int i = idx(p, r);
          if (i < 0) continue;
          R.g_red[i] += gg[(size_t)d * 6 + r]; R.g_cam[i] += gc[(size_t)d * 6 + r]; R.Udiag[i] += Ud[(size_t)d * 6 + r];
        }
      }
      for (int u = 0; u < nu; ++u) {
        const int pa = b.dev_to_pose[b.ui[u]], pb = b.dev_to_pose[b.uj[u]];
        const double* src = Ho + (size_t)u * 36;
        if (pa <= pb) { auto& blk = R.H[{pa, pb}]; for (int k = 0; k < 36; ++k) blk[k] += src[k]; }
        else { auto& blk = R.H[{pb, pa}]; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) blk[r * 6 + c] += src[c * 6 + r]; }
      }
    }
  };
  // back-substitutes the points for the (unscaled) camera step; out3 += [model decrease, |dX|^2, |X|^2]
  auto bundle_step = [&](const std::vector<double>& step, double* out3) {
    StageTimer stage_timer_bs_("solve: point back-substitution / candidate cost (GPU)");
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      const int nd = (int)b.dev_to_pose.size();
      std::vector<double> dcam((size_t)nd * 6, 0.0);
      for (int d = 0; d < nd; ++d)
        for (int r = 0; r < 6; ++r) { const int i = idx(b.dev_to_pose[d], r); if (i >= 0) dcam[(size_t)d * 6 + r] = step[i]; }
      double o[3];
      e.Check(pvlm_ba_step(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, dcam.data(), o), "pvlm_ba_step");
      for (int k = 0; k < 3; ++k) out3[k] += o[k];
    }
  };
  auto bundle_cost = [&](const std::vector<double>& v, bool candidate) {
    StageTimer stage_timer_bc_("solve: point back-substitution / candidate cost (GPU)");
    double c = 0.0;
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      bundle_poses(b, v);
      double ci = 0.0;
      e.Check(pvlm_ba_cost(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, candidate ? 1 : 0, &ci), "pvlm_ba_cost");
      c += ci;
    }
    return c;
  };
  auto finish_points = [&]() {   // the refined structure goes back into the caller's point blocks
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      std::vector<double> X(b.point_blocks.size() * 3);
      e.Check(pvlm_ba_get_points(e.ctx(), b.set, 0, X.data()), "pvlm_ba_get_points");
      for (size_t p = 0; p < b.point_blocks.size(); ++p) for (int k = 0; k < 3; ++k) I.blocks[b.point_blocks[p]][k] = X[3 * p + k];
    }
  };

  double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
  Assembled A;
  evaluate(x, true, A);
  Reduced R; R.g_red.assign(n_free, 0.0); R.g_cam.assign(n_free, 0.0); R.Udiag.assign(n_free, 0.0);
  bool R_valid = true;
  if (have_bundles) bundle_reduce(x, radius, true, R);
  double cost = A.cost + R.cost;
  summary->initial_cost = summary->final_cost = cost;
  summary->cost_history.push_back(cost);
  summary->num_successful_steps = 1;  // iteration 0 counts as successful in Ceres' summary ([recalled])
  summary->usable = std::isfinite(cost);
  if (!summary->usable) { summary->message = "initial cost is not finite"; return; }
  if (n_free == 0 && !have_bundles) return;

  // envelope of the camera/LiDAR system (four-block groups + Schur complement blocks)
  auto build_first = [&](const Assembled& As, const Reduced& Rs) {
    std::vector<int> first(n_free);
    for (int i = 0; i < n_free; ++i) first[i] = i;
    auto add = [&](const std::pair<int, int>& key) {
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const int i = idx(key.first, r), j = idx(key.second, c);
        if (i < 0 || j < 0) continue;
        const int hi = std::max(i, j), lo = std::min(i, j);
        first[hi] = std::min(first[hi], lo);
      }
    };
    ForEachBlock(As, [&](const std::pair<int, int>& key, const double*) { add(key); });
    ForEachBlock(Rs.H, [&](const std::pair<int, int>& key, const double*) { add(key); });
    return first;
  };
  // diagonal of the FULL J^T J on the free pose columns (before any elimination)
  auto full_diag = [&](const Assembled& As, const Reduced& Rs) {
    std::vector<double> d(n_free, 0.0);
    ForEachBlock(As, [&](const std::pair<int, int>& key, const double* blk) {
      if (key.first != key.second) return;
      for (int r = 0; r < 6; ++r) { const int i = idx(key.first, r); if (i >= 0) d[i] += blk[r * 6 + r]; }   // += : two poses may share a parameter block
    });
    for (int i = 0; i < n_free; ++i) d[i] += Rs.Udiag[i];
    return d;
  };

  // Jacobi scaling from the initial Jacobian: 1 / (1 + sqrt(diag(J^T J)))   (Ceres jacobi_scaling)
  std::vector<double> scale(n_free, 1.0);
  {
    const std::vector<double> d0 = full_diag(A, R);
    for (int i = 0; i < n_free; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(std::max(0.0, d0[i])));
  }

  int iter = 0;
  auto gmax = [&](const Assembled& As, const Reduced& Rs) {
    double m = Rs.gmax_points;
    for (int i = 0; i < n_free; ++i) m = std::max(m, std::fabs(As.g[i] + Rs.g_cam[i]));
    return m;
  };
  if (gmax(A, R) <= opt.gradient_tolerance) { summary->message = "gradient tolerance reached"; finish_points(); return; }
  auto fill = [&](Skyline& S, const auto& H) {
    ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const int i = idx(key.first, r), j = idx(key.second, c);
        if (i < 0 || j < 0) continue;
        const double v = blk[r * 6 + c] * scale[i] * scale[j];
        if (key.first == key.second) { if (i >= j) S.at(i, j) += v; }   // diagonal block: lower triangle once
        else if (i >= j) S.at(i, j) += v; else S.at(j, i) += v;
      }
    });
  };
  // Large reduced systems (Room / Floor sized joint problems: thousands of unknowns) are assembled, factorised and solved
  // on the GPU (pvlm_spd_solve_blocks: blocked Cholesky kernels); small ones by the host skyline Cholesky.
  const bool gpu_chol = n_free >= gpu_chol_min;
  // v^T (D H D) v over a block list, without forming the matrix
  auto quad_form = [&](const auto& H, const std::vector<double>& v) {
    double q = 0.0;
    ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
      double b = 0.0;
      for (int r = 0; r < 6; ++r) {
        const int i = idx(key.first, r);
        if (i < 0) continue;
        double row = 0.0;
        for (int c = 0; c < 6; ++c) { const int j = idx(key.second, c); if (j >= 0) row += blk[r * 6 + c] * scale[j] * v[j]; }
        b += scale[i] * v[i] * row;
      }
      q += key.first == key.second ? b : 2.0 * b;
    });
    return q;
  };
  std::vector<int> chol_rows, chol_cols, chol_mirror; std::vector<double> chol_blocks;     // block lists of the GPU Cholesky (see below)
  unsigned long long chol_key_hash = 0;
  while (iter < opt.max_num_iterations) {
    ++iter;
    if (!R_valid) { bundle_reduce(x, radius, false, R); R_valid = true; }
    // scaled system  (D (H + S_points) D + diag(clamp(diag(D J^T J D))) / radius) dy = -D g
    const std::vector<double> dfull = full_diag(A, R);
    std::vector<double> rhs(n_free), damp(n_free);
    for (int i = 0; i < n_free; ++i) {
      rhs[i] = -(A.g[i] + R.g_red[i]) * scale[i];
      const double hs = dfull[i] * scale[i] * scale[i];
      damp[i] = std::min(std::max(hs, opt.min_lm_diagonal), opt.max_lm_diagonal) / radius;
    }
    bool step_ok;
    std::vector<double> dy = rhs;
    double model_change = 0.0, dn = 0.0, xn = 0.0;
    Skyline S0;                      // four-block groups only: the Gauss-Newton model of those blocks
    if (gpu_chol) {
      StageTimer stage_timer_chol_("solve: GPU Cholesky");
      StageTimer* stage_timer_push_ = new StageTimer("  (inside the GPU Cholesky stage) host block list");
      // the index lists depend on the structure only: made by the first step of the Solve, checked (block count) and reused by the others — only the values move
      std::vector<int>& rows = chol_rows; std::vector<int>& cols = chol_cols; std::vector<int>& mirror = chol_mirror; std::vector<double>& blocks = chol_blocks;
      size_t n_pushed = 0;
      unsigned long long key_hash = 1469598103934665603ull;          // of the block keys in push order: the reused lists must belong to exactly this sequence
      const bool have_lists = !mirror.empty();
      blocks.clear();
      auto push = [&](const auto& H) {
        ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
          if (!have_lists) {
            for (int r = 0; r < 6; ++r) { rows.push_back(idx(key.first, r)); cols.push_back(idx(key.second, r)); }
            mirror.push_back(key.first != key.second ? 1 : 0);
          }
          blocks.insert(blocks.end(), blk, blk + 36);
          key_hash = (key_hash ^ (unsigned long long)(unsigned)key.first) * 1099511628211ull;
          key_hash = (key_hash ^ (unsigned long long)(unsigned)key.second) * 1099511628211ull;
          ++n_pushed;
        });
      };
      push(A);
      if (have_bundles) push(R.H);
      if (!have_lists) chol_key_hash = key_hash;
      if (have_lists && (n_pushed != mirror.size() || key_hash != chol_key_hash)) {               // never seen (the structure of a Solve is fixed); rebuilt rather than trusted
        rows.clear(); cols.clear(); mirror.clear(); blocks.clear(); n_pushed = 0;
        auto push_all = [&](const auto& H) {
          ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
            for (int r = 0; r < 6; ++r) { rows.push_back(idx(key.first, r)); cols.push_back(idx(key.second, r)); }
            mirror.push_back(key.first != key.second ? 1 : 0);
            blocks.insert(blocks.end(), blk, blk + 36);
          });
        };
        push_all(A);
        if (have_bundles) push_all(R.H);
        chol_key_hash = key_hash;
      }
      delete stage_timer_push_;
      int info = 0;
      long long hits_before = 0, hits_after = 0;
      pvlm_spd_plan_prefetch_hits(e.ctx(), &hits_before);
      e.Check(pvlm_spd_solve_blocks(e.ctx(), n_free, (int)mirror.size(), rows.data(), cols.data(), mirror.data(), blocks.data(), scale.data(), damp.data(),
                                    dy.data(), &info), "pvlm_spd_solve_blocks");
      pvlm_spd_plan_prefetch_hits(e.ctx(), &hits_after);
      if (hits_after > hits_before) { StageTimer stage_timer_hit_("  (inside the GPU Cholesky stage) plan taken from the prefetch started on entry to the Solve (count)"); }
      step_ok = info == 0;
    } else {
      S0.Init(build_first(A, R));
      fill(S0, A);
      Skyline S = S0;
      if (have_bundles) fill(S, R.H);
      for (int i = 0; i < n_free; ++i) S.at(i, i) += damp[i];
      { StageTimer stage_timer_chol_("solve: host skyline Cholesky"); step_ok = n_free == 0 || S.Factor(); }
      if (step_ok && n_free) S.Solve(dy);
    }
    if (step_ok) {
      // model_cost_change = -(g'.dy + 1/2 dy^T H' dy) over the four-block groups ...
      double gd = 0.0, dHd = 0.0;
      for (int i = 0; i < n_free; ++i) gd += (A.g[i] * scale[i]) * dy[i];
      if (gpu_chol) {
        std::vector<double> unit(n_free);
        for (int i = 0; i < n_free; ++i) unit[i] = dy[i];
        dHd = quad_form(A, unit);
      } else {
        for (int i = 0; i < n_free; ++i) {
          double s = 0.0;
          for (int k = S0.first[i]; k < i; ++k) s += S0.at(i, k) * dy[k];
          dHd += dy[i] * (2.0 * s + S0.at(i, i) * dy[i]);
        }
      }
      model_change = -(gd + 0.5 * dHd);
      // ... plus the reprojection blocks' own model decrease after back-substituting their points
      if (have_bundles) {
        std::vector<double> step(n_free);
        for (int i = 0; i < n_free; ++i) step[i] = dy[i] * scale[i];
        double o3[3] = {0, 0, 0};
        bundle_step(step, o3);
        model_change += o3[0]; dn += o3[1]; xn += o3[2];
      }
      step_ok = model_change > 0.0 && std::isfinite(model_change);
    }
    bool accepted = false;
    if (step_ok) {
      std::vector<double> cand = x;
      for (size_t b = 0; b < I.blocks.size(); ++b)
        if (block_off[b] >= 0) for (int k = 0; k < 3; ++k) { const double d = dy[block_off[b] + k] * scale[block_off[b] + k]; cand[3 * b + k] += d; dn += d * d; xn += x[3 * b + k] * x[3 * b + k]; }
      Assembled C;
      evaluate(cand, true, C);
      const double ccost = C.cost + (have_bundles ? bundle_cost(cand, true) : 0.0);
      const double rho = (cost - ccost) / model_change;
      if (opt.minimizer_progress_to_stdout)
        printf("iter %2d cost %.8e -> %.8e  model %.3e rho %.3f radius %.3e\n", iter, cost, ccost, model_change, rho, radius);
      if (std::isfinite(ccost) && rho > opt.min_relative_decrease) {
        accepted = true;
        const double cost_change = cost - ccost;
        x = cand; A = std::move(C);
        for (auto& b : I.bundles) if (b.set) e.Check(pvlm_ba_accept(e.ctx(), b.set), "pvlm_ba_accept");
        const double f = 1.0 - std::pow(2.0 * rho - 1.0, 3);
        radius = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, f));
        decrease_factor = 2.0;
        if (have_bundles) { bundle_reduce(x, radius, false, R); R_valid = true; }   // gradient at the new point + next system
        summary->num_successful_steps++;
        const double prev = cost;
        cost = ccost;
        summary->cost_history.push_back(cost);
        if (std::fabs(cost_change) <= opt.function_tolerance * prev) { summary->message = "function tolerance reached"; break; }
        if (gmax(A, R) <= opt.gradient_tolerance) { summary->message = "gradient tolerance reached"; break; }
        if (std::sqrt(dn) <= opt.parameter_tolerance * (std::sqrt(xn) + opt.parameter_tolerance)) { summary->message = "parameter tolerance reached"; break; }
      }
    }
    if (!accepted) {
      summary->num_unsuccessful_steps++;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      R_valid = !have_bundles;
      if (radius < opt.min_trust_region_radius) { summary->message = "trust region collapsed"; break; }
    }
  }
  if (summary->message.empty()) summary->message = "maximum number of iterations reached";
  store_x(x);
  finish_points();
  summary->final_cost = cost;
  summary->usable = std::isfinite(cost);
}

}  // namespace ceres_like

// ================================================================================================
// functor factories — base/CostFunction.h ::Create
// ================================================================================================
using ceres_like::CostFunction;
static CostFunction* MakeCost(int kind, unsigned flags, double weight, std::initializer_list<double> row) {
  CostFunction* c = new CostFunction();
  c->kind = kind; c->flags = flags; c->weight = weight; c->row.assign(row.begin(), row.end());
  return c;
}
CostFunction* Point2Plane_Meter::Create(const Vector3d& p, const Vector4d& pl, const double w) {
  return MakeCost(PVLM_POINT2PLANE_METER, 0, w, {p[0], p[1], p[2], pl[0], pl[1], pl[2], pl[3]});
}
CostFunction* Point2Plane_Angle::Create(const Vector3d& p, const Vector4d& pl, const bool normalize, const double w) {
  return MakeCost(PVLM_POINT2PLANE_ANGLE, normalize ? PVLM_FLAG_NORMALIZE_DISTANCE : 0, w, {p[0], p[1], p[2], pl[0], pl[1], pl[2], pl[3]});
}
CostFunction* Point2Line_Meter::Create(const Vector3d& p, const Vector3d& a, const Vector3d& b, const double w) {
  return MakeCost(PVLM_POINT2LINE_METER, 0, w, {p[0], p[1], p[2], a[0], a[1], a[2], b[0], b[1], b[2]});
}
CostFunction* Point2Line_Angle::Create(const Vector3d& p, const Vector3d& a, const Vector3d& b, const bool normalize, const double w) {
  return MakeCost(PVLM_POINT2LINE_ANGLE, normalize ? PVLM_FLAG_NORMALIZE_DISTANCE : 0, w, {p[0], p[1], p[2], a[0], a[1], a[2], b[0], b[1], b[2]});
}
CostFunction* Plane2Plane_Global::Create(const Vector3d& n, const Vector3d& a, const Vector3d& b, const double w) {
  return MakeCost(PVLM_PLANE2PLANE_GLOBAL, 0, 1.0, {n[0], n[1], n[2], a[0], a[1], a[2], b[0], b[1], b[2], w});
}
CostFunction* PanoramaReprojResidual_1Angle::Create(const Vector3d& pt, double w) {
  CostFunction* c = MakeCost(kReprojKind, 0, w, {pt[0], pt[1], pt[2]});
  c->num_blocks = 3;
  return c;
}
CostFunction* PlaneIOUResidual::Create(const Vector4d& pl, const Vector3d& mn, const Vector3d& mr, const double angle, const double w) {
  return MakeCost(PVLM_PLANE_IOU, 0, 1.0, {pl[0], pl[1], pl[2], pl[3], mn[0], mn[1], mn[2], mr[0], mr[1], mr[2], angle, w});
}

// ================================================================================================
// Exchange factories
// ================================================================================================
Exchange MakeRcclExchange(int world, int rank, const unsigned char id[128]) {
  Engine& e = Engine::Default();
  struct State {
    pvlm_comm* comm = nullptr;
    ~State() { if (comm) pvlm_comm_destroy(Engine::Default().ctx(), comm); }
  };
  auto st = std::make_shared<State>();
  e.Check(pvlm_comm_create(e.ctx(), world, rank, id, &st->comm), "pvlm_comm_create");
  Exchange x; x.world = world; x.rank = rank;
  x.allreduce_sum = [st](double* buf, size_t count) {
    Engine& en = Engine::Default();
    en.Check(pvlm_allreduce_sum_f64_host(en.ctx(), st->comm, buf, (int64_t)count), "pvlm_allreduce_sum_f64_host");
  };
  return x;
}

std::pair<size_t, size_t> Exchange::BalancedRange(const std::vector<double>& weight, int of_rank) const {
  const size_t n = weight.size();
  const size_t rk = (size_t)(of_rank < 0 ? rank : of_rank), W = (size_t)std::max(world, 1);
  double total = 0;
  for (double w : weight) total += w;
  if (!(total > 0)) return {n * rk / W, n * (rk + 1) / W};
  auto boundary = [&](size_t r) -> size_t {
    if (r == 0) return 0;
    if (r >= W) return n;
    const double want = total * (double)r / (double)W;
    double acc = 0;
    for (size_t i = 0; i < n; ++i) { if (acc >= want) return i; acc += weight[i]; }
    return n;
  };
  return {boundary(rk), boundary(rk + 1)};
}

Exchange MakeFileExchange(int world, int rank, const std::string& dir) {
  auto seq = std::make_shared<long>(0);
  // The ranks of a file exchange are processes on ONE machine that usually share ONE GPU (the functional check of the sharded path where no second GPU exists): the
  // pose solve by level launches for them — the one-launch form's workgroups wait for each other and want the GPU's workgroup slots to themselves (pvlm_spd_one_launch),
  // and every rank has to take the same form for the ranks to agree bit for bit.  One process per GPU (MakeRcclExchange) keeps the default.
  if (world > 1) { Engine& en = Engine::Default(); en.Check(pvlm_spd_one_launch(en.ctx(), 0, nullptr), "pvlm_spd_one_launch"); }
  Exchange x; x.world = world; x.rank = rank;
  x.allreduce_sum = [world, rank, dir, seq](double* buf, size_t count) {
    const long s = (*seq)++;
    auto name = [&](long q, int r) { return dir + "/x" + std::to_string(q) + "_" + std::to_string(r) + ".bin"; };
    {
      const std::string tmp = name(s, rank) + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f) throw std::runtime_error("file exchange: cannot write " + tmp);
      const uint64_t n = count;
      fwrite(&n, sizeof(n), 1, f); fwrite(buf, sizeof(double), count, f);
      fclose(f);
      if (rename(tmp.c_str(), name(s, rank).c_str()) != 0) throw std::runtime_error("file exchange: rename failed");
    }
    std::vector<double> sum(count, 0.0), part(count);
    for (int r = 0; r < world; ++r) {             // rank order: the same sum, bit for bit, on every rank
      FILE* f = nullptr;
      for (int spin = 0; spin < 1200000 && !(f = fopen(name(s, r).c_str(), "rb")); ++spin) std::this_thread::sleep_for(std::chrono::microseconds(100));
      if (!f) throw std::runtime_error("file exchange: rank " + std::to_string(r) + " never arrived at exchange " + std::to_string(s));
      uint64_t n = 0;
      if (fread(&n, sizeof(n), 1, f) != 1 || n != count || fread(part.data(), sizeof(double), count, f) != count) {
        fclose(f);
        throw std::runtime_error("file exchange: size mismatch between ranks (exchange " + std::to_string(s) + ": rank " + std::to_string(r) + " wrote " + std::to_string((unsigned long long)n) +
                                 " doubles, rank " + std::to_string(rank) + " expects " + std::to_string((unsigned long long)count) + ")");
      }
      fclose(f);
      for (size_t i = 0; i < count; ++i) sum[i] += part[i];
    }
    std::copy(sum.begin(), sum.end(), buf);
    if (s >= 2) remove(name(s - 2, rank).c_str());   // every rank has read exchange s-2 before it wrote s-1, and all of s-1 has been read here
  };
  return x;
}


}  // namespace pvlm
