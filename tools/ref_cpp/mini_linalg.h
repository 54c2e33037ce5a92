// mini_linalg.h -- the few vector / matrix members that the extracted reference function bodies use.
// TEST-FIXTURE TOOLING ONLY (tools/make_ref_cpp_fixtures.py); never part of the product or the oracle.
//
// The reference's leaf functions (contact law, PGS sweep, Butcher tableaux) are written against Eigen 3.4, which
// is not installed in this image.  This header gives the generated translation unit just enough of the same
// SPELLING (`Eigen::Vector3d::dot`, `A.col(i).dot(x)`, `x.segment(o, n)`, `(MatrixXd(r, c) << ...).finished()`)
// for those bodies to compile unchanged.  It is a stand-in for a missing third-party header: fixtures produced
// with it are labelled tier "B" (reference TEXT executed, not a reference BUILD) in the .npz and in DESIGN.md.
// Every operation below is the plain scalar definition, evaluated left to right in double precision; there is
// no vectorisation and no fused multiply-add (the TU is compiled with -ffp-contract=off), which is also what
// Eigen's scalar path does for these sizes up to summation order inside `dot` (sequential here; Eigen's
// redux for dynamic vectors may pair-sum: differences of that origin are bounded by a few ulp and are why the
// tier-B comparisons use 1e-14 / 1e-15 relative instead of bit equality).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen
{
using Index = std::ptrdiff_t;
constexpr int Infinity = -1;

// ---------------------------------------------------------------- fixed 3-vector
struct Vector3d
{
    double d[3] = {0.0, 0.0, 0.0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : d{x, y, z} {}
    static Vector3d Zero() { return Vector3d{}; }
    double & operator[](Index i) { return d[i]; }
    double operator[](Index i) const { return d[i]; }
    double dot(const Vector3d & o) const { return d[0] * o.d[0] + d[1] * o.d[1] + d[2] * o.d[2]; }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    Vector3d & noalias() { return *this; }
    void setZero() { d[0] = d[1] = d[2] = 0.0; }
    Vector3d & operator-=(const Vector3d & o) { for (int i = 0; i < 3; ++i) d[i] -= o.d[i]; return *this; }
    Vector3d & operator+=(const Vector3d & o) { for (int i = 0; i < 3; ++i) d[i] += o.d[i]; return *this; }
    Vector3d & operator*=(double s) { for (int i = 0; i < 3; ++i) d[i] *= s; return *this; }
};
inline Vector3d operator*(double s, const Vector3d & v) { return {s * v.d[0], s * v.d[1], s * v.d[2]}; }
inline Vector3d operator*(const Vector3d & v, double s) { return {v.d[0] * s, v.d[1] * s, v.d[2] * s}; }
inline Vector3d operator-(const Vector3d & a, const Vector3d & b) { return {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}; }
inline Vector3d operator+(const Vector3d & a, const Vector3d & b) { return {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}; }

// ---------------------------------------------------------------- dynamic vector + views
struct BoolArray
{
    std::vector<char> v;
    bool all() const { return std::all_of(v.begin(), v.end(), [](char c) { return c != 0; }); }
};
struct AbsArray
{
    std::vector<double> v;
    AbsArray abs() const { AbsArray r; for (double x : v) r.v.push_back(std::fabs(x)); return r; }
    AbsArray array() const { return *this; }
    BoolArray operator<(double t) const { BoolArray r; for (double x : v) r.v.push_back(x < t); return r; }
};

struct VectorSegment            // `VectorXd::SegmentReturnType`: a window on somebody else's storage
{
    double * p = nullptr;
    Index n = 0;
    double & operator[](Index i) const { return p[i]; }
    Index size() const { return n; }
    VectorSegment segment(Index o, Index m) const { return {p + o, m}; }
};

struct VectorXd
{
    using SegmentReturnType = VectorSegment;
    std::vector<double> v;
    VectorXd() = default;
    explicit VectorXd(Index n) : v(static_cast<size_t>(n), 0.0) {}
    void resize(Index n) { v.assign(static_cast<size_t>(n), 0.0); }
    Index size() const { return static_cast<Index>(v.size()); }
    double & operator[](Index i) { return v[static_cast<size_t>(i)]; }
    double operator[](Index i) const { return v[static_cast<size_t>(i)]; }
    void setZero() { std::fill(v.begin(), v.end(), 0.0); }
    VectorSegment segment(Index o, Index m) { return {v.data() + o, m}; }
    VectorSegment head(Index m) { return {v.data(), m}; }
    template<int P> double lpNorm() const
    {
        static_assert(P == Infinity, "only the infinity norm is provided");
        double m = 0.0;
        for (double x : v) m = std::max(m, std::fabs(x));
        return m;
    }
    AbsArray operator-(const VectorXd & o) const
    {
        AbsArray r;
        for (size_t i = 0; i < v.size(); ++i) r.v.push_back(v[i] - o.v[i]);
        return r;
    }
    // `(VectorXd(n) << a, b, c).finished()`
    struct Comma
    {
        VectorXd * self;
        size_t k;
        Comma & operator,(double x) { self->v[k++] = x; return *this; }
        VectorXd finished() { return *self; }
    };
    Comma operator<<(double x) { v[0] = x; return {this, 1}; }
};

// ---------------------------------------------------------------- dynamic matrix, column major like Eigen's default
struct MatrixColumn
{
    const double * p;
    Index n;
    template<typename Vec> double dot(const Vec & x) const
    {
        double s = 0.0;
        for (Index k = 0; k < n; ++k) s += p[k] * x[k];
        return s;
    }
};
struct MatrixXd
{
    std::vector<double> v;
    Index r = 0, c = 0;
    MatrixXd() = default;
    MatrixXd(Index rows, Index cols) : v(static_cast<size_t>(rows * cols), 0.0), r(rows), c(cols) {}
    Index rows() const { return r; }
    Index cols() const { return c; }
    double & operator()(Index i, Index j) { return v[static_cast<size_t>(j * r + i)]; }
    double operator()(Index i, Index j) const { return v[static_cast<size_t>(j * r + i)]; }
    MatrixColumn col(Index j) const { return {v.data() + j * r, r}; }
    // `(MatrixXd(r, c) << ...).finished()`: the comma initialiser fills ROW by ROW
    struct Comma
    {
        MatrixXd * self;
        Index k;
        Comma & operator,(double x) { (*self)(k / self->c, k % self->c) = x; ++k; return *this; }
        MatrixXd finished() { return *self; }
    };
    Comma operator<<(double x) { (*this)(0, 0) = x; return {this, 1}; }
};
}  // namespace Eigen

namespace pinocchio
{
// `pinocchio::Force{linear, angular}`: a pair of 3-vectors, nothing else is used by the extracted bodies
struct Force
{
    Eigen::Vector3d lin, ang;
    Force(const Eigen::Vector3d & l, const Eigen::Vector3d & a) : lin(l), ang(a) {}
    const Eigen::Vector3d & linear() const { return lin; }
    const Eigen::Vector3d & angular() const { return ang; }
};
}  // namespace pinocchio
