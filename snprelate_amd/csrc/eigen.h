// Internal interface of the top-k eigen solver (eigen.hip) shared with the multi-device driver (multi.hip).
#pragma once
#include <vector>

#include "snpgpu_internal.h"

namespace snpgpu {

// y = scale * C q for blocks of vectors stored vector-major: Q, Y = double [b][n] on device(); apply() overwrites Y and
// returns when Y is complete.  fp32_products: the solver accepts a product with relative error ~1e-7 (kernels_eig.hip)
struct EigOperator {
    virtual ~EigOperator() {}
    virtual int64_t n() const = 0;
    virtual int device() const = 0;
    virtual int apply(const double *Q, int b, double *Y, bool fp32_products) = 0;
    virtual bool has_fp32_products() const { return true; }
};

// row panels resident on ONE device (a whole matrix, or one rank's share of it: then `reduce` sums `y_buf` over the ranks)
class PanelsOperator : public EigOperator {
public:
    PanelsOperator(const std::vector<snpgpu_ctx *> &panels, int64_t n, int device, double scale, double *y_buf,
                   snpgpu_reduce_fn reduce, void *user)
        : panels_(panels), n_(n), dev_(device), scale_(scale), y_buf_(y_buf), reduce_(reduce), user_(user) {}
    int64_t n() const override { return n_; }
    int device() const override { return dev_; }
    int apply(const double *Q, int b, double *Y, bool fp32_products) override;
    bool has_fp32_products() const override { return getenv("SNPGPU_EIG_BLAS") == nullptr; }      // not the two-dgemm form

private:
    std::vector<snpgpu_ctx *> panels_;
    int64_t n_;
    int dev_;
    double scale_;
    double *y_buf_;
    snpgpu_reduce_fn reduce_;
    void *user_;
};

// largest-k eigenpairs: eigval_host [k] descending (host), eigvec [k][n] = column-major n x k (`mem`: host or op.device())
int krylov_topk(EigOperator &op, int k, const snpgpu_eig_opts *opts, double *eigval_host, double *eigvec, int mem,
                snpgpu_eig_info *info);

}  // namespace snpgpu
