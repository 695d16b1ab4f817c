// densecrf stand-in (TEST INFRASTRUCTURE ONLY).  densecrf is a third-party dependency that is NOT part of the reference tree
// (martinruenz/densecrf fork, HEAD clone, Scripts/install.sh:84).  The reference drives the inference loop itself
// (Segmentation.cpp:452-470: expAndNormalize, getPotential(k)->apply); this header supplies those two operations from the ORACLE
// (oracle/orc_segment.c: exact Gaussian kernels with symmetric normalisation, blocked summation order) in place of the
// permutohedral-lattice approximation, so that everything the reference does around them runs on the reference's own code.
#pragma once
#include <Eigen/Core>
#include <memory>
#include <vector>
extern "C" void orc_crf_kernel(const float* feat, int D, int n, float* Kn);
extern "C" void orc_crf_apply(const float* Kn, int n, int L, float w, const float* Q, float* out);
extern "C" void orc_crf_exp_and_normalize(const float* in, float* out, int L, int n);
enum KernelType { CONST_KERNEL, DIAG_KERNEL, FULL_KERNEL };
enum NormalizationType { NO_NORMALIZATION, NORMALIZE_BEFORE, NORMALIZE_AFTER, NORMALIZE_SYMMETRIC };
class LabelCompatibility { public: virtual ~LabelCompatibility() {} virtual float weight() const = 0; };
class PottsCompatibility : public LabelCompatibility { public: explicit PottsCompatibility(float w = 1.f) : w_(w) {} float weight() const override { return w_; } private: float w_; };
class PairwisePotential {
  public:
    PairwisePotential(const Eigen::MatrixXf& features, LabelCompatibility* compat) : n_(features.cols()), K_((size_t)features.cols() * features.cols()), compat_(compat)
    {   // features: D x n column-major == [n][D] row-major
        orc_crf_kernel(features.data(), features.rows(), n_, K_.data());
    }
    // out = compatibility(kernel * Q): Potts => -w * (K Q)   (Q: L x n column-major == [n][L] row-major)
    void apply(Eigen::MatrixXf& out, const Eigen::MatrixXf& Q) const
    {
        out.resize(Q.rows(), Q.cols());
        orc_crf_apply(K_.data(), n_, Q.rows(), compat_->weight(), Q.data(), out.data());
    }
  private:
    int n_;
    std::vector<float> K_;
    std::unique_ptr<LabelCompatibility> compat_;
};
class DenseCRF {
  public:
    static void expAndNormalize(Eigen::MatrixXf& out, const Eigen::MatrixXf& in)
    {
        Eigen::MatrixXf tmp(in.rows(), in.cols());  // `in` may alias `out`
        orc_crf_exp_and_normalize(in.data(), tmp.data(), in.rows(), in.cols());
        out = tmp;
    }
};
class DenseCRF2D : public DenseCRF {
  public:
    DenseCRF2D(int W, int H, int M) : W_(W), H_(H), M_(M) {}
    void setUnaryEnergy(const Eigen::MatrixXf&) {}  // the reference's own loop reads `unary` directly (Segmentation.cpp:458-470)
    void addPairwiseGaussian(float sx, float sy, LabelCompatibility* c, KernelType = DIAG_KERNEL, NormalizationType = NORMALIZE_SYMMETRIC)
    {   // densecrf: feature = (x / sx, y / sy) per pixel
        Eigen::MatrixXf f(2, W_ * H_);
        for (int j = 0; j < H_; j++) for (int i = 0; i < W_; i++) { f(0, j * W_ + i) = (float)i / sx; f(1, j * W_ + i) = (float)j / sy; }
        pots_.emplace_back(new PairwisePotential(f, c));
    }
    void addPairwiseEnergy(const Eigen::MatrixXf& features, LabelCompatibility* c, KernelType = DIAG_KERNEL, NormalizationType = NORMALIZE_SYMMETRIC)
    {
        pots_.emplace_back(new PairwisePotential(features, c));
    }
    unsigned countPotentials() const { return (unsigned)pots_.size(); }
    PairwisePotential* getPotential(unsigned k) { return pots_[k].get(); }
  private:
    int W_, H_, M_;
    std::vector<std::unique_ptr<PairwisePotential>> pots_;
};
