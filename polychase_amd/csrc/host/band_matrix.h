// band_matrix.h -- symmetric positive definite band matrix with an in-place Cholesky (fp64: the reference factorises in fp32, which
// loses the low-frequency modes of long chains).  J^T J of
// the refiner is block-banded (a frame only shares residuals with frames a few steps away), so this
// replaces Eigen::SimplicialLLT of the reference (cpp/pnp/lev_marq.h:394-397, :826-842).
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

// lower band stored row by row: row r holds columns r - bw ... r
class BandMatrix {
   public:
    BandMatrix(int n, int half_bandwidth) : n_(n), bw_(half_bandwidth), v_(static_cast<size_t>(n) * (half_bandwidth + 1), 0.0) {}
    int Size() const { return n_; }
    double& At(int r, int c) { return v_[static_cast<size_t>(r) * (bw_ + 1) + (c - r + bw_)]; }  // r - bw <= c <= r
    double At(int r, int c) const { return v_[static_cast<size_t>(r) * (bw_ + 1) + (c - r + bw_)]; }
    void SetZero() { std::fill(v_.begin(), v_.end(), 0.0); }

    // y = A x with A symmetric (selfadjointView<Lower>)
    void Multiply(const std::vector<double>& x, std::vector<double>& y) const {
        y.assign(static_cast<size_t>(n_), 0.0);
        for (int r = 0; r < n_; r++) {
            double s = At(r, r) * x[r];
            for (int c = std::max(0, r - bw_); c < r; c++) {
                const double a = At(r, c);
                s += a * x[c];
                y[c] += a * x[r];
            }
            y[r] += s;
        }
    }
    // in-place Cholesky A = L L^T; false if a pivot is not positive
    bool Factorize() {
        for (int r = 0; r < n_; r++) {
            const int c0 = std::max(0, r - bw_);
            for (int c = c0; c <= r; c++) {
                // row r and row c are contiguous in k: four independent partial sums (the single chain was bound by the
                // latency of the multiply-add: 0.8 ms per factorisation of the 1800 x 1800, half bandwidth 47 system of C5)
                const int k0 = std::max(c0, c - bw_), len = c - k0;
                const double* a = &At(r, k0);
                const double* b = &At(c, k0);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int k = 0;
                for (; k + 4 <= len; k += 4) {
                    s0 += a[k] * b[k];
                    s1 += a[k + 1] * b[k + 1];
                    s2 += a[k + 2] * b[k + 2];
                    s3 += a[k + 3] * b[k + 3];
                }
                for (; k < len; k++) s0 += a[k] * b[k];
                const double s = At(r, c) - ((s0 + s1) + (s2 + s3));
                if (c < r) {
                    At(r, c) = s / At(c, c);
                } else {
                    if (!(s > 0.0) || !std::isfinite(s)) return false;
                    At(r, r) = std::sqrt(s);
                }
            }
        }
        return true;
    }
    // x = (L L^T)^-1 b
    void Solve(const std::vector<double>& b, std::vector<double>& x) const {
        x = b;
        for (int r = 0; r < n_; r++) {
            double s = x[r];
            for (int c = std::max(0, r - bw_); c < r; c++) s -= At(r, c) * x[c];
            x[r] = s / At(r, r);
        }
        for (int r = n_ - 1; r >= 0; r--) {
            const double xr = x[r] / At(r, r);
            x[r] = xr;
            for (int c = std::max(0, r - bw_); c < r; c++) x[c] -= At(r, c) * xr;
        }
    }

   private:
    int n_, bw_;
    std::vector<double> v_;
};

