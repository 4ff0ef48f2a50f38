// INTERFACE-ONLY TEST DOUBLE — THIS IS NOT CERES.
//
// The few classes of the public Ceres 2.0 API that integration/pvlm_ceres.hpp touches, declared with the signatures
// recalled from Ceres 2.0.0's headers (cost_function.h, sized_cost_function.h, evaluation_callback.h, loss_function.h,
// problem.h) so that the adapter can be COMPILED and its rows CHECKED in an image that has no Ceres.  It implements no
// solver and pins nothing about Ceres' behaviour: Problem only stores the blocks it is given and can replay the calls a
// Ceres evaluation makes (PrepareForEvaluation once, then every block's Evaluate).  Never shipped, never linked by the
// product.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <set>
#include <vector>

namespace ceres {

class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }
 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }
 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_ = 0;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() { set_num_residuals(kNumResiduals); *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...}; }
};

class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = a_ / r; rho[2] = -rho[1] / (2.0 * s); }
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
 private:
  double a_, b_;
};

class Problem {
 public:
  struct Options { EvaluationCallback* evaluation_callback = nullptr; };
  Problem() {}
  explicit Problem(const Options& o) : options_(o) {}
  ~Problem() { for (auto& b : blocks_) if (costs_.insert(b.cost).second) delete b.cost; for (LossFunction* l : losses_) delete l; }
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1, double* x2, double* x3) {
    blocks_.push_back(Block{cost, loss, {x0, x1, x2, x3}, 4});
    if (loss) losses_.insert(loss);
    return &blocks_.back();
  }
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1, double* x2) {
    blocks_.push_back(Block{cost, loss, {x0, x1, x2, nullptr}, 3});
    if (loss) losses_.insert(loss);
    return &blocks_.back();
  }
  int NumResidualBlocks() const { return (int)blocks_.size(); }
  // test-double only: what one Ceres evaluation does — the callback once, then every block (residual + one Jacobian row of up
  // to 12 entries, 3 per parameter block, at J[12 i ...]; a three-block cost fills the first 9)
  void EvaluateAll(bool jacobians, bool new_point, std::vector<double>* r, std::vector<double>* J) {
    if (options_.evaluation_callback) options_.evaluation_callback->PrepareForEvaluation(jacobians, new_point);
    r->assign(blocks_.size(), 0.0);
    if (J) J->assign(blocks_.size() * 12, 0.0);
    for (size_t i = 0; i < blocks_.size(); ++i) {
      double* jac[4] = {nullptr, nullptr, nullptr, nullptr};
      if (jacobians && J) for (int b = 0; b < blocks_[i].n; ++b) jac[b] = J->data() + 12 * i + 3 * b;
      blocks_[i].cost->Evaluate(blocks_[i].x, &(*r)[i], jacobians && J ? jac : nullptr);
    }
  }
  // the loss a block was added with (nullptr = none) and its parameter blocks: what the adapter tests check against the reference's calls
  const LossFunction* loss_of(size_t i) const { return blocks_[i].loss; }
  double* const* parameters_of(size_t i) const { return blocks_[i].x; }
  int num_parameter_blocks_of(size_t i) const { return blocks_[i].n; }
 private:
  struct Block { CostFunction* cost; LossFunction* loss; double* x[4]; int n; };
  Options options_;
  std::vector<Block> blocks_;
  std::set<CostFunction*> costs_;
  std::set<LossFunction*> losses_;
};

}  // namespace ceres
