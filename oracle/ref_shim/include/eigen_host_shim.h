// TEST INFRASTRUCTURE (oracle/): a dense stand-in for the handful of Eigen types the reference's VSLAM/backend/src/gn_kernels.cu uses
// on the HOST side of its Gauss-Newton solvers (class SparseBlock, gn_kernels.cu:56-157): SparseMatrix<double> (setFromTriplets with
// duplicate summation, copy, A - B, diagonal().array() += ep + lm * diagonal().array()), Triplet<double>, VectorX<double> / VectorXd
// (Zero, operator(), a - b, data()), MatrixXd(SparseMatrix) (column-major data()), SimplicialLLT (compute / info / solve), Success.
// Eigen itself is not in this image (VSLAM/setup.py:18-21 takes it from thirdparty/eigen, an absent submodule).  The systems are
// 7 x (keyframes - 1) unknowns, so a dense fp64 Cholesky (LL^T, the factorisation SimplicialLLT computes) is exact for the purpose;
// like SimplicialLLT it reports failure on a non-positive pivot.  Never part of the product.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {

enum ComputationInfo { Success = 0, NumericalIssue = 1 };

template <class S> struct Triplet {
    int r, c; S v;
    Triplet(int r_, int c_, S v_) : r(r_), c(c_), v(v_) {}
    int row() const { return r; }
    int col() const { return c; }
    S value() const { return v; }
};

template <class S> class VectorX {
public:
    std::vector<S> d;
    VectorX() {}
    explicit VectorX(size_t n) : d(n, S(0)) {}
    static VectorX Zero(size_t n) { return VectorX(n); }
    S& operator()(size_t i) { return d[i]; }
    const S& operator()(size_t i) const { return d[i]; }
    size_t size() const { return d.size(); }
    S* data() { return d.data(); }
    const S* data() const { return d.data(); }
    VectorX operator-(const VectorX& o) const { VectorX r(d.size()); for (size_t i = 0; i < d.size(); ++i) r.d[i] = d[i] - o.d[i]; return r; }
};
typedef VectorX<double> VectorXd;

// values of a diagonal expression (`ep + lm * L.diagonal().array()`)
struct DiagVals { std::vector<double> v; };
inline DiagVals operator+(double a, DiagVals x) { for (auto& e : x.v) e = a + e; return x; }
inline DiagVals operator+(DiagVals x, double a) { for (auto& e : x.v) e = e + a; return x; }

template <class S> class SparseMatrix;
template <class S> struct DiagArrayRef {
    SparseMatrix<S>* m;
    DiagVals values() const;
    DiagArrayRef& operator+=(const DiagVals& x);
};
template <class S> inline DiagVals operator*(double a, const DiagArrayRef<S>& d) { DiagVals x = d.values(); for (auto& e : x.v) e = a * e; return x; }
template <class S> struct DiagRef { SparseMatrix<S>* m; DiagArrayRef<S> array() { return DiagArrayRef<S>{m}; } };

template <class S> class SparseMatrix {   // stored dense, row-major
public:
    size_t n_r = 0, n_c = 0;
    std::vector<S> d;
    SparseMatrix() {}
    SparseMatrix(size_t r, size_t c) : n_r(r), n_c(c), d(r * c, S(0)) {}
    size_t rows() const { return n_r; }
    size_t cols() const { return n_c; }
    S& at(size_t r, size_t c) { return d[r * n_c + c]; }
    const S& at(size_t r, size_t c) const { return d[r * n_c + c]; }
    template <class It> void setFromTriplets(It b, It e) {   // replaces the content; duplicates are summed (Eigen's contract)
        std::fill(d.begin(), d.end(), S(0));
        for (It it = b; it != e; ++it) at(it->row(), it->col()) += it->value();
    }
    SparseMatrix operator-(const SparseMatrix& o) const { SparseMatrix r(n_r, n_c); for (size_t i = 0; i < d.size(); ++i) r.d[i] = d[i] - o.d[i]; return r; }
    DiagRef<S> diagonal() { return DiagRef<S>{this}; }
};
template <class S> DiagVals DiagArrayRef<S>::values() const { DiagVals x; for (size_t i = 0; i < m->n_r; ++i) x.v.push_back((double)m->at(i, i)); return x; }
template <class S> DiagArrayRef<S>& DiagArrayRef<S>::operator+=(const DiagVals& x) { for (size_t i = 0; i < m->n_r; ++i) m->at(i, i) += (S)x.v[i]; return *this; }

class MatrixXd {   // column-major like Eigen's default
public:
    size_t n_r = 0, n_c = 0;
    std::vector<double> d;
    MatrixXd() {}
    explicit MatrixXd(const SparseMatrix<double>& a) : n_r(a.rows()), n_c(a.cols()), d(a.rows() * a.cols()) {
        for (size_t r = 0; r < n_r; ++r) for (size_t c = 0; c < n_c; ++c) d[c * n_r + r] = a.at(r, c);
    }
    double* data() { return d.data(); }
};

template <class M> class SimplicialLLT {   // dense LL^T of the (symmetric) matrix, lower triangle
    std::vector<double> L;
    size_t n = 0;
    ComputationInfo st = NumericalIssue;
public:
    void compute(const M& a) {
        n = a.rows();
        L.assign(n * n, 0.0);
        st = Success;
        for (size_t j = 0; j < n; ++j) {
            double s = a.at(j, j);
            for (size_t k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
            if (!(s > 0.0)) { st = NumericalIssue; return; }
            const double ljj = std::sqrt(s);
            L[j * n + j] = ljj;
            for (size_t i = j + 1; i < n; ++i) {
                double t = a.at(i, j);
                for (size_t k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
                L[i * n + j] = t / ljj;
            }
        }
    }
    ComputationInfo info() const { return st; }
    VectorXd solve(const VectorXd& b) const {
        VectorXd y(n), x(n);
        for (size_t i = 0; i < n; ++i) { double s = b(i); for (size_t k = 0; k < i; ++k) s -= L[i * n + k] * y(k); y(i) = s / L[i * n + i]; }
        for (size_t ii = n; ii-- > 0;) { double s = y(ii); for (size_t k = ii + 1; k < n; ++k) s -= L[k * n + ii] * x(k); x(ii) = s / L[ii * n + ii]; }
        return x;
    }
};

} // namespace Eigen
