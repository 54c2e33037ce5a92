// jm_math.h -- per-lane spatial algebra for the gfx950 kernels (one robot per wavefront lane).
//
// Everything here operates on values held in VGPRs of ONE lane; there is no cross-lane traffic.
// Conventions follow the reference engine's numerics layer (Pinocchio v2.7.0 semantics, see
// SURVEY.md appendix A): spatial vectors are [linear; angular], SE3 (R, p) maps child
// coordinates to the parent.  The articulated inertia is kept in symmetric block form
// (A = lin-lin 6 scalars, B = lin-ang 9 scalars, D = ang-ang 6 scalars) instead of the dense
// 6x6 the reference stores (pinocchio_overload_algorithms.h:151 `data.Yaba`).
#pragma once
#ifdef JM_HOST_EMU
// Host emulation build (tests only): the same per-lane code compiled by g++ so that the kernel
// logic can be checked against the oracle on machines without a GPU. Never part of the product.
#include <cmath>
#define JM_DEV inline
#define JM_REFRESH() ((void)0)
#define JM_OPAQUE(x) ((void)0)
#define JM_OPAQUE_S(x) ((void)0)
#define JM_K(c, kz) (c)
#define JM_KZ(dep) 0
#else
#include <hip/hip_runtime.h>
#define JM_DEV __device__ __forceinline__
// Compiler-only memory barrier: values read from the (lane-uniform or LDS) constant tables must
// be re-read after it instead of being hoisted out of the evaluation loop / kept live across the
// ABA sweeps -- re-reading LDS or the scalar cache is far cheaper than the VGPRs it would pin.
// Opaque re-definition of a per-lane integer: address arithmetic that depends on it cannot be
// hoisted out of the evaluation loop (LICM otherwise pins one 64-bit VGPR address per store).
#define JM_OPAQUE(x) asm volatile("" : "+v"(x))
// the same for a wave-uniform value (scalar register): re-defining the parameter-block pointer inside the evaluation
// loop keeps its scalar loads inside the loop -- hoisted, they overflow the SGPR file and come back as v_readlane
#define JM_OPAQUE_S(x) asm volatile("" : "+s"(x))
// Literal constant `c` made data-dependent on the (loop-variant) value `dep`: the constant is materialised
// into a scalar register pair next to its use (2 SALU moves) instead of being hoisted out of the evaluation
// loop into a VGPR pair by LICM -- where the polynomial coefficients of sincos / tanh ended up in scratch as
// soon as the kernel is built for two waves per SIMD (256 VGPRs).
#define JM_KZ(dep) jm::kzero_(dep)
#define JM_K(c, kz) __hiloint2double((int)(__builtin_bit_cast(unsigned long long, (double)(c)) >> 32) | (kz), (int)__builtin_bit_cast(unsigned long long, (double)(c)) | (kz))
#ifndef JM_NO_REFRESH
#define JM_REFRESH() asm volatile("" ::: "memory")
#else
#define JM_REFRESH() ((void)0)
#endif
#endif

namespace jm
{
#ifndef JM_FP_REASSOC
#define JM_FP_REASSOC 1   // (0: A/B builds; measured in round 4, same box: ANYmal launch 0.1517 -> 0.1480 ms, Atlas 0.3606 -> 0.3477)
#endif
#if JM_FP_REASSOC && !defined(JM_HOST_EMU)
// The spatial-algebra helpers below (and only they: the pragma is switched off again in front of the scalar functions, whose
// Cody-Waite reductions and polynomials depend on their order of operations): a sum of products may be re-associated into one
// fma chain, `a + (x y + z w + u v)` -> three fmas instead of mul + 2 fma + add
#pragma clang fp reassociate(on)
#endif
#ifndef JM_HOST_EMU
// a uniform zero the compiler cannot see through, defined after `dep` is (see JM_K)
__device__ __forceinline__ int kzero_(double dep) { int z; asm("s_mov_b32 %0, 0" : "=s"(z) : "v"(dep)); return z; }
#endif
template<class T> struct V3
{
    T x, y, z;
};
template<class T> JM_DEV V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template<class T> JM_DEV V3<T> zero3() { return V3<T>{T(0), T(0), T(0)}; }
template<class T> JM_DEV V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template<class T> JM_DEV V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template<class T> JM_DEV V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template<class T> JM_DEV V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template<class T> JM_DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template<class T> JM_DEV V3<T> cross(V3<T> a, V3<T> b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template<class T> JM_DEV T comp(V3<T> a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

// general 3x3, row major
template<class T> struct M3
{
    T m00, m01, m02, m10, m11, m12, m20, m21, m22;
};
template<class T> JM_DEV M3<T> ident3() { return {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}; }
template<class T> JM_DEV V3<T> operator*(const M3<T> & A, V3<T> v)
{
    return {A.m00 * v.x + A.m01 * v.y + A.m02 * v.z,
            A.m10 * v.x + A.m11 * v.y + A.m12 * v.z,
            A.m20 * v.x + A.m21 * v.y + A.m22 * v.z};
}
template<class T> JM_DEV V3<T> tmul(const M3<T> & A, V3<T> v)  // A^T v
{
    return {A.m00 * v.x + A.m10 * v.y + A.m20 * v.z,
            A.m01 * v.x + A.m11 * v.y + A.m21 * v.z,
            A.m02 * v.x + A.m12 * v.y + A.m22 * v.z};
}
template<class T> JM_DEV M3<T> operator*(const M3<T> & A, const M3<T> & B)
{
    return {A.m00 * B.m00 + A.m01 * B.m10 + A.m02 * B.m20, A.m00 * B.m01 + A.m01 * B.m11 + A.m02 * B.m21, A.m00 * B.m02 + A.m01 * B.m12 + A.m02 * B.m22,
            A.m10 * B.m00 + A.m11 * B.m10 + A.m12 * B.m20, A.m10 * B.m01 + A.m11 * B.m11 + A.m12 * B.m21, A.m10 * B.m02 + A.m11 * B.m12 + A.m12 * B.m22,
            A.m20 * B.m00 + A.m21 * B.m10 + A.m22 * B.m20, A.m20 * B.m01 + A.m21 * B.m11 + A.m22 * B.m21, A.m20 * B.m02 + A.m21 * B.m12 + A.m22 * B.m22};
}
template<class T> JM_DEV M3<T> mul_bt(const M3<T> & A, const M3<T> & B)  // A B^T
{
    return {A.m00 * B.m00 + A.m01 * B.m01 + A.m02 * B.m02, A.m00 * B.m10 + A.m01 * B.m11 + A.m02 * B.m12, A.m00 * B.m20 + A.m01 * B.m21 + A.m02 * B.m22,
            A.m10 * B.m00 + A.m11 * B.m01 + A.m12 * B.m02, A.m10 * B.m10 + A.m11 * B.m11 + A.m12 * B.m12, A.m10 * B.m20 + A.m11 * B.m21 + A.m12 * B.m22,
            A.m20 * B.m00 + A.m21 * B.m01 + A.m22 * B.m02, A.m20 * B.m10 + A.m21 * B.m11 + A.m22 * B.m12, A.m20 * B.m20 + A.m21 * B.m21 + A.m22 * B.m22};
}
template<class T> JM_DEV M3<T> transpose(const M3<T> & A)
{
    return {A.m00, A.m10, A.m20, A.m01, A.m11, A.m21, A.m02, A.m12, A.m22};
}
template<class T> JM_DEV M3<T> operator+(const M3<T> & A, const M3<T> & B)
{
    return {A.m00 + B.m00, A.m01 + B.m01, A.m02 + B.m02, A.m10 + B.m10, A.m11 + B.m11, A.m12 + B.m12, A.m20 + B.m20, A.m21 + B.m21, A.m22 + B.m22};
}
template<class T> JM_DEV M3<T> operator-(const M3<T> & A, const M3<T> & B)
{
    return {A.m00 - B.m00, A.m01 - B.m01, A.m02 - B.m02, A.m10 - B.m10, A.m11 - B.m11, A.m12 - B.m12, A.m20 - B.m20, A.m21 - B.m21, A.m22 - B.m22};
}
// p^ A  (skew(p) times A)
template<class T> JM_DEV M3<T> skew_mul(V3<T> p, const M3<T> & A)
{
    return {p.y * A.m20 - p.z * A.m10, p.y * A.m21 - p.z * A.m11, p.y * A.m22 - p.z * A.m12,
            p.z * A.m00 - p.x * A.m20, p.z * A.m01 - p.x * A.m21, p.z * A.m02 - p.x * A.m22,
            p.x * A.m10 - p.y * A.m00, p.x * A.m11 - p.y * A.m01, p.x * A.m12 - p.y * A.m02};
}
// A p^  (A times skew(p))
template<class T> JM_DEV M3<T> mul_skew(const M3<T> & A, V3<T> p)
{
    return {A.m01 * p.z - A.m02 * p.y, A.m02 * p.x - A.m00 * p.z, A.m00 * p.y - A.m01 * p.x,
            A.m11 * p.z - A.m12 * p.y, A.m12 * p.x - A.m10 * p.z, A.m10 * p.y - A.m11 * p.x,
            A.m21 * p.z - A.m22 * p.y, A.m22 * p.x - A.m20 * p.z, A.m20 * p.y - A.m21 * p.x};
}

// symmetric 3x3
template<class T> struct S3
{
    T xx, xy, xz, yy, yz, zz;
};
template<class T> JM_DEV V3<T> operator*(const S3<T> & A, V3<T> v)
{
    return {A.xx * v.x + A.xy * v.y + A.xz * v.z, A.xy * v.x + A.yy * v.y + A.yz * v.z, A.xz * v.x + A.yz * v.y + A.zz * v.z};
}
template<class T> JM_DEV S3<T> operator+(const S3<T> & A, const S3<T> & B)
{
    return {A.xx + B.xx, A.xy + B.xy, A.xz + B.xz, A.yy + B.yy, A.yz + B.yz, A.zz + B.zz};
}
template<class T> JM_DEV M3<T> full(const S3<T> & A) { return {A.xx, A.xy, A.xz, A.xy, A.yy, A.yz, A.xz, A.yz, A.zz}; }
// R A R^T for symmetric A
template<class T> JM_DEV S3<T> rot_sym(const M3<T> & R, const S3<T> & A)
{
    // L = R A
    const T l00 = R.m00 * A.xx + R.m01 * A.xy + R.m02 * A.xz, l01 = R.m00 * A.xy + R.m01 * A.yy + R.m02 * A.yz, l02 = R.m00 * A.xz + R.m01 * A.yz + R.m02 * A.zz;
    const T l10 = R.m10 * A.xx + R.m11 * A.xy + R.m12 * A.xz, l11 = R.m10 * A.xy + R.m11 * A.yy + R.m12 * A.yz, l12 = R.m10 * A.xz + R.m11 * A.yz + R.m12 * A.zz;
    const T l20 = R.m20 * A.xx + R.m21 * A.xy + R.m22 * A.xz, l21 = R.m20 * A.xy + R.m21 * A.yy + R.m22 * A.yz, l22 = R.m20 * A.xz + R.m21 * A.yz + R.m22 * A.zz;
    return {l00 * R.m00 + l01 * R.m01 + l02 * R.m02, l00 * R.m10 + l01 * R.m11 + l02 * R.m12, l00 * R.m20 + l01 * R.m21 + l02 * R.m22,
            l10 * R.m10 + l11 * R.m11 + l12 * R.m12, l10 * R.m20 + l11 * R.m21 + l12 * R.m22,
            l20 * R.m20 + l21 * R.m21 + l22 * R.m22};
}
// symmetric part helper: S = M + M^T
template<class T> JM_DEV S3<T> sym_of(const M3<T> & M)
{
    return {M.m00 + M.m00, M.m01 + M.m10, M.m02 + M.m20, M.m11 + M.m11, M.m12 + M.m21, M.m22 + M.m22};
}
// -p^ A p^ = p^ A p^T for symmetric A, result symmetric
template<class T> JM_DEV S3<T> skew_sym_skewT(V3<T> p, const S3<T> & A)
{
    const M3<T> PA = skew_mul(p, full(A));  // p^ A
    // (p^ A) p^T = -(p^ A) p^
    const M3<T> Q = mul_skew(PA, p);
    return {-Q.m00, -Q.m01, -Q.m02, -Q.m11, -Q.m12, -Q.m22};
}

template<class T> struct SE3
{
    M3<T> R;
    V3<T> p;
};
template<class T> JM_DEV SE3<T> operator*(const SE3<T> & A, const SE3<T> & B) { return {A.R * B.R, A.p + A.R * B.p}; }

template<class T> struct Sp  // spatial motion or force, [linear; angular]
{
    V3<T> l, a;
};
template<class T> JM_DEV Sp<T> zero6() { return {zero3<T>(), zero3<T>()}; }
template<class T> JM_DEV Sp<T> operator+(Sp<T> a, Sp<T> b) { return {a.l + b.l, a.a + b.a}; }
template<class T> JM_DEV Sp<T> operator-(Sp<T> a, Sp<T> b) { return {a.l - b.l, a.a - b.a}; }
template<class T> JM_DEV Sp<T> act_motion(const SE3<T> & M, Sp<T> m)
{
    const V3<T> Rw = M.R * m.a;
    return {M.R * m.l + cross(M.p, Rw), Rw};
}
template<class T> JM_DEV Sp<T> actinv_motion(const SE3<T> & M, Sp<T> m)
{
    return {tmul(M.R, m.l - cross(M.p, m.a)), tmul(M.R, m.a)};
}
template<class T> JM_DEV Sp<T> act_force(const SE3<T> & M, Sp<T> f)
{
    const V3<T> Rf = M.R * f.l;
    return {Rf, M.R * f.a + cross(M.p, Rf)};
}
template<class T> JM_DEV Sp<T> actinv_force(const SE3<T> & M, Sp<T> f)
{
    return {tmul(M.R, f.l), tmul(M.R, f.a - cross(M.p, f.l))};
}
template<class T> JM_DEV Sp<T> cross_mm(Sp<T> a, Sp<T> b)  // motion x motion
{
    return {cross(a.a, b.l) + cross(a.l, b.a), cross(a.a, b.a)};
}
template<class T> JM_DEV Sp<T> cross_mf(Sp<T> a, Sp<T> f)  // motion x* force
{
    return {cross(a.a, f.l), cross(a.a, f.a) + cross(a.l, f.l)};
}

// rigid-body inertia (mass, com, rotational inertia about the com)
template<class T> struct RBI
{
    T m;
    V3<T> c;
    S3<T> I;
};
template<class T> JM_DEV Sp<T> rbi_mul(const RBI<T> & Y, Sp<T> v)
{
    const V3<T> l = Y.m * (v.l - cross(Y.c, v.a));
    return {l, Y.I * v.a + cross(Y.c, l)};
}
template<class T> JM_DEV T rbi_vtiv(const RBI<T> & Y, Sp<T> v)
{
    const V3<T> d = v.l - cross(Y.c, v.a);
    return Y.m * dot(d, d) + dot(v.a, Y.I * v.a);
}

// articulated inertia in symmetric block form:  [A  B; B^T  D]
template<class T> struct AI
{
    S3<T> A;
    M3<T> B;
    S3<T> D;
};
template<class T> JM_DEV AI<T> ai_from_rbi(const RBI<T> & Y)
{
    const V3<T> c = Y.c;
    const T m = Y.m;
    AI<T> r;
    r.A = {m, T(0), T(0), m, T(0), m};
    // B = -m c^  ( = m c^T )
    r.B = {T(0), m * c.z, -m * c.y, -m * c.z, T(0), m * c.x, m * c.y, -m * c.x, T(0)};
    // D = I - m c^ c^ ;  c^ c^ = c c^T - |c|^2 1
    const T cc = dot(c, c);
    r.D = {Y.I.xx + m * (cc - c.x * c.x), Y.I.xy - m * c.x * c.y, Y.I.xz - m * c.x * c.z,
           Y.I.yy + m * (cc - c.y * c.y), Y.I.yz - m * c.y * c.z, Y.I.zz + m * (cc - c.z * c.z)};
    return r;
}
template<class T> JM_DEV Sp<T> ai_mul(const AI<T> & Y, Sp<T> v)  // force = Y * motion
{
    return {Y.A * v.l + Y.B * v.a, tmul(Y.B, v.l) + Y.D * v.a};
}
template<class T> JM_DEV AI<T> operator+(const AI<T> & X, const AI<T> & Y) { return {X.A + Y.A, X.B + Y.B, X.D + Y.D}; }
// Y -= (1/d) U U^T  with U = [ul; ua]
template<class T> JM_DEV void ai_rank1_sub(AI<T> & Y, Sp<T> U, T dinv)
{
    const V3<T> sl = dinv * U.l, sa = dinv * U.a;
    Y.A.xx -= sl.x * U.l.x; Y.A.xy -= sl.x * U.l.y; Y.A.xz -= sl.x * U.l.z;
    Y.A.yy -= sl.y * U.l.y; Y.A.yz -= sl.y * U.l.z; Y.A.zz -= sl.z * U.l.z;
    Y.B.m00 -= sl.x * U.a.x; Y.B.m01 -= sl.x * U.a.y; Y.B.m02 -= sl.x * U.a.z;
    Y.B.m10 -= sl.y * U.a.x; Y.B.m11 -= sl.y * U.a.y; Y.B.m12 -= sl.y * U.a.z;
    Y.B.m20 -= sl.z * U.a.x; Y.B.m21 -= sl.z * U.a.y; Y.B.m22 -= sl.z * U.a.z;
    Y.D.xx -= sa.x * U.a.x; Y.D.xy -= sa.x * U.a.y; Y.D.xz -= sa.x * U.a.z;
    Y.D.yy -= sa.y * U.a.y; Y.D.yz -= sa.y * U.a.z; Y.D.zz -= sa.z * U.a.z;
}
// X_f Y X_f^T : express an articulated inertia of the child frame in the parent frame
// (internal::SE3actOn, pinocchio_overload_algorithms.h:163-164)
template<class T> JM_DEV AI<T> ai_transform(const SE3<T> & M, const AI<T> & Y)
{
    AI<T> r;
    r.A = rot_sym(M.R, Y.A);
    const M3<T> Br = mul_bt(M.R * Y.B, M.R);  // R B R^T
    const S3<T> Dr = rot_sym(M.R, Y.D);
    r.B = Br - mul_skew(full(r.A), M.p);  // Br - A' p^
    // D' = Dr + p^ Br + (p^ Br)^T - p^ A' p^
    const M3<T> PB = skew_mul(M.p, Br);
    r.D = Dr + sym_of(PB) + skew_sym_skewT(M.p, r.A);
    // sym_of(PB) = PB + PB^T  (diagonal doubled as required)
    return r;
}

// rotation helpers
template<class T> JM_DEV M3<T> rot_axis(int k, T c, T s)  // about +x (0), +y (1), +z (2)
{
    if (k == 0) return {T(1), T(0), T(0), T(0), c, -s, T(0), s, c};
    if (k == 1) return {c, T(0), s, T(0), T(1), T(0), -s, T(0), c};
    return {c, -s, T(0), s, c, T(0), T(0), T(0), T(1)};
}
template<class T> JM_DEV M3<T> rot_rodrigues(V3<T> a, T c, T s)
{
    const T oc = T(1) - c;
    return {c + oc * a.x * a.x, oc * a.x * a.y - s * a.z, oc * a.x * a.z + s * a.y,
            oc * a.y * a.x + s * a.z, c + oc * a.y * a.y, oc * a.y * a.z - s * a.x,
            oc * a.z * a.x - s * a.y, oc * a.z * a.y + s * a.x, c + oc * a.z * a.z};
}
template<class T> JM_DEV M3<T> quat_to_matrix(T x, T y, T z, T w)
{
    const T tx = T(2) * x, ty = T(2) * y, tz = T(2) * z;
    const T twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x;
    const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    return {T(1) - (tyy + tzz), txy - twz, txz + twy,
            txy + twz, T(1) - (txx + tzz), tyz - twx,
            txz - twy, tyz + twx, T(1) - (txx + tyy)};
}

#if JM_FP_REASSOC && !defined(JM_HOST_EMU)
#pragma clang fp reassociate(off)
#endif
// sin and cos of a float64 angle without the libm `sincos(x, &s, &c)` out-pointer form: on the
// device that form materialises its outputs through private (scratch) memory, and hipcc (ROCm 7.2)
// was observed to lay those slots over live spill slots in the large unrolled kernels (wrong
// accelerations after an integrate step, GPU only).  Cody-Waite reduction by pi/2 in three FMA
// steps + the fdlibm kernel polynomials (< 2 ulp for |x| < 1e5; larger arguments give NaN).
JM_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a * b + k with the constant k (JM_K) read straight from its scalar register pair (VOP3 form; the compiler's own
// choice is `v_fmac` with the addend copied into a VGPR pair first: two more VALU moves per coefficient)
#ifdef JM_HOST_EMU
JM_DEV double fmak_(double a, double b, double k) { return __builtin_fma(a, b, k); }
#else
JM_DEV double fmak_(double a, double b, double k)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}
#endif
JM_DEV void sincos_(double x, double * s, double * c)
{
    // Branch-free: |x| >= 1e5 is never a valid joint angle or dt * omega; it yields NaN, which
    // flags the lane (JM_LANE_NAN) instead of taking a Payne-Hanek slow path that would split
    // every kinematics block of the kernels in two.
    x = (__builtin_fabs(x) < 1.0e5) ? x : __builtin_nan("");
    const double fn = __builtin_rint(x * 0.63661977236758134308);
    double r = fma_(-fn, 1.5707963267948966, x);
    r = fma_(-fn, 6.123233995736766e-17, r);
    r = fma_(-fn, -1.4973849048591698e-33, r);
    const double z = r * r;
    const int kz = JM_KZ(z);
    (void)kz;
    // Horner chains with the addends in scalar registers (JM_K): `v_fma_f64 d, z, p, s[c]`
    double ps = fmak_(z, JM_K(1.58969099521155010221e-10, kz), JM_K(-2.50507602534068634195e-08, kz));
    ps = fmak_(z, ps, JM_K(2.75573137070700676789e-06, kz));
    ps = fmak_(z, ps, JM_K(-1.98412698298579493134e-04, kz));
    ps = fmak_(z, ps, JM_K(8.33333333332248946124e-03, kz));
    const double sk = fma_(z * r, fmak_(z, ps, JM_K(-1.66666666666666324348e-01, kz)), r);
    double pc = fmak_(z, JM_K(-1.13596475577881948265e-11, kz), JM_K(2.08757232129817482790e-09, kz));
    pc = fmak_(z, pc, JM_K(-2.75573143513906633035e-07, kz));
    pc = fmak_(z, pc, JM_K(2.48015872894767294178e-05, kz));
    pc = fmak_(z, pc, JM_K(-1.38888888888741095749e-03, kz));
    pc = fmak_(z, pc, JM_K(4.16666666666666019037e-02, kz));
    pc = z * pc;
    const double ck = 1.0 - (0.5 * z - z * pc);
    const int n = (int)fn & 3;
    const double s0 = (n & 1) ? ck : sk, c0 = (n & 1) ? sk : ck;
    *s = (n & 2) ? -s0 : s0;
    *c = ((n + 1) & 2) ? -c0 : c0;
}
#ifdef JM_HOST_EMU
JM_DEV void sincos_(float x, float * s, float * c) { *s = std::sin(x); *c = std::cos(x); }
#else
JM_DEV void sincos_(float x, float * s, float * c) { *s = ::sinf(x); *c = ::cosf(x); }
#endif
// reciprocal of a well-scaled positive number (joint-space inertias D, LDL^T pivots): hardware
// estimate + two Newton steps (<= 1 ulp) instead of the 12-instruction IEEE division sequence
// (no denormal / overflow rescaling: D is a physical inertia, never near the exponent limits)
#ifdef JM_HOST_EMU
JM_DEV double rcp_(double x) { return 1.0 / x; }
JM_DEV float rcp_(float x) { return 1.0f / x; }
#else
JM_DEV double rcp_(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}
JM_DEV float rcp_(float x) { return 1.0f / x; }
#endif
JM_DEV double sqrt_(double x) { return ::sqrt(x); }
JM_DEV float sqrt_(float x) { return ::sqrtf(x); }
// 1 / sqrt(x) of a well-scaled positive number: hardware estimate + two Newton steps
#ifdef JM_HOST_EMU
JM_DEV double rsqrt_(double x) { return 1.0 / ::sqrt(x); }
#else
JM_DEV double rsqrt_(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(y, __builtin_fma(-0.5 * x * y, y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-0.5 * x * y, y, 0.5), y);
    return y;
}
#endif
JM_DEV float rsqrt_(float x) { return 1.0f / ::sqrtf(x); }
JM_DEV double trunc_(double x) { return ::trunc(x); }
JM_DEV float trunc_(float x) { return ::truncf(x); }
// tanh(x) = e / (e + 2), e = expm1(2 |x|) = 2^n (expm1(r) + 1) - 1 with 2|x| = n ln2 + r, |r| <= ln2 / 2 (degree-13
// Taylor polynomial of expm1, truncation 4e-18 relative); |x| >= 20 saturates.  <= 2 ulp; replaces ocml's
// extended-precision tanh (3x the instructions, a dozen coefficients hoisted into VGPRs by LICM).
JM_DEV double tanh_(double x)
{
    const double ax = __builtin_fabs(x);
    const double y = __builtin_fmin(ax + ax, 40.0);
    const double fn = __builtin_rint(y * 1.4426950408889634074);
    double r = fma_(-fn, 6.93147180369123816490e-01, y);
    r = fma_(-fn, 1.90821492927058770002e-10, r);
    const int kz = JM_KZ(r);
    double p = fmak_(r, JM_K(1.0 / 6227020800.0, kz), JM_K(1.0 / 479001600.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 39916800.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 3628800.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 362880.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 40320.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 5040.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 720.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 120.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 24.0, kz));
    p = fmak_(r, p, JM_K(1.0 / 6.0, kz));
    p = fma_(r, p, 0.5);
    p = fma_(r * r, p, r);                        // expm1(r)
    const double s = __builtin_ldexp(1.0, (int)fn);   // 2^n, n in 0..58
    const double e = fma_(s, p, s - 1.0);        // expm1(y)
    const double t = e * rcp_(e + 2.0);
    return __builtin_copysign(x != x ? x : t, x);
}
JM_DEV float tanh_(float x) { return ::tanhf(x); }
JM_DEV double atan2_(double y, double x) { return ::atan2(y, x); }
JM_DEV float atan2_(float y, float x) { return ::atan2f(y, x); }
JM_DEV double fmin_(double a, double b) { return ::fmin(a, b); }
JM_DEV float fmin_(float a, float b) { return ::fminf(a, b); }
JM_DEV double fmax_(double a, double b) { return ::fmax(a, b); }
JM_DEV float fmax_(float a, float b) { return ::fmaxf(a, b); }
template<class T> JM_DEV T clamp_(T x, T lo, T hi) { return fmin_(fmax_(x, lo), hi); }
template<class T> struct Eps;
template<> struct Eps<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double taylor = 1.220703125e-4; };
template<> struct Eps<float> { static constexpr float eps = 1.1920929e-7f; static constexpr float taylor = 1.8581361e-2f; };

// exp6 of a twist [v; w] (SE(3) exponential; Taylor expansion below eps^(1/4))
template<class T> JM_DEV SE3<T> exp6(Sp<T> nu)
{
    const V3<T> v = nu.l, w = nu.a;
    const T t2 = dot(w, w);
    const T t = sqrt_(t2);
    T st, ct;
    sincos_(t, &st, &ct);
    const bool small = t < Eps<T>::taylor;
    // (reciprocals by rcp_: the IEEE division sequence is 2-3x the instructions, and both sides of every select
    // below are evaluated)
    const T inv_t = rcp_(small ? T(1) : t);
    const T inv_t2 = inv_t * inv_t;
    const T a_wxv = small ? T(0.5) - t2 * T(1.0 / 24.0) : (T(1) - ct) * inv_t2;
    const T a_v = small ? T(1) - t2 * T(1.0 / 6.0) : st * inv_t;
    const T a_w = small ? T(1.0 / 6.0) - t2 * T(1.0 / 120.0) : (T(1) - a_v) * inv_t2;
    const T dg = small ? T(1) - t2 * T(0.5) : ct;
    SE3<T> M;
    M.p = a_v * v + (a_w * dot(w, v)) * w + a_wxv * cross(w, v);
    M.R = {a_wxv * w.x * w.x + dg, a_wxv * w.x * w.y - a_v * w.z, a_wxv * w.x * w.z + a_v * w.y,
           a_wxv * w.y * w.x + a_v * w.z, a_wxv * w.y * w.y + dg, a_wxv * w.y * w.z - a_v * w.x,
           a_wxv * w.z * w.x - a_v * w.y, a_wxv * w.z * w.y + a_v * w.x, a_wxv * w.z * w.z + dg};
    return M;
}
// rotation matrix -> quaternion xyzw (Eigen's branch structure)
template<class T> JM_DEV void matrix_to_quat(const M3<T> & R, T & x, T & y, T & z, T & w)
{
    T t = R.m00 + R.m11 + R.m22;
    if (t > T(0))
    {
        t = sqrt_(t + T(1));
        w = T(0.5) * t;
        t = T(0.5) * rcp_(t);
        x = (R.m21 - R.m12) * t;
        y = (R.m02 - R.m20) * t;
        z = (R.m10 - R.m01) * t;
    }
    else if (R.m00 >= R.m11 && R.m00 >= R.m22)
    {
        t = sqrt_(R.m00 - R.m11 - R.m22 + T(1));
        x = T(0.5) * t;
        t = T(0.5) * rcp_(t);
        w = (R.m21 - R.m12) * t;
        y = (R.m10 + R.m01) * t;
        z = (R.m20 + R.m02) * t;
    }
    else if (R.m11 >= R.m22)
    {
        t = sqrt_(R.m11 - R.m22 - R.m00 + T(1));
        y = T(0.5) * t;
        t = T(0.5) * rcp_(t);
        w = (R.m02 - R.m20) * t;
        z = (R.m21 + R.m12) * t;
        x = (R.m01 + R.m10) * t;
    }
    else
    {
        t = sqrt_(R.m22 - R.m00 - R.m11 + T(1));
        z = T(0.5) * t;
        t = T(0.5) * rcp_(t);
        w = (R.m10 - R.m01) * t;
        x = (R.m02 + R.m20) * t;
        y = (R.m12 + R.m21) * t;
    }
}
// ---- SO(3) logarithm (adaptive error norm: jm_adaptive.h; orientation error of a user FrameConstraint: jm_constraint.h)
JM_DEV double acos_(double x) { return ::acos(x); }
JM_DEV float acos_(float x) { return ::acosf(x); }
JM_DEV double asin_(double x) { return ::asin(x); }
JM_DEV float asin_(float x) { return ::asinf(x); }
JM_DEV double fabs_(double x) { return ::fabs(x); }
JM_DEV float fabs_(float x) { return ::fabsf(x); }

// Pinocchio v2.7.0 log3
template<class T> JM_DEV V3<T> log3(const M3<T> & R)
{
    const T PI_value = T(3.14159265358979323846);
    T tr = R.m00 + R.m11 + R.m22;
    T theta;
    if (tr >= T(3)) { tr = T(3); theta = T(0); }
    else if (tr <= T(-1)) { tr = T(-1); theta = PI_value; }
    else theta = acos_((tr - T(1)) / T(2));
    if (theta >= PI_value - T(1e-2))
    {
        const T cphi = -(tr - T(1)) / T(2);
        const T beta = theta * theta / (T(1) + cphi);
        const T t0 = (R.m00 + cphi) * beta, t1 = (R.m11 + cphi) * beta, t2 = (R.m22 + cphi) * beta;
        return {(R.m21 > R.m12 ? T(1) : T(-1)) * (t0 > T(0) ? sqrt_(t0) : T(0)),
                (R.m02 > R.m20 ? T(1) : T(-1)) * (t1 > T(0) ? sqrt_(t1) : T(0)),
                (R.m10 > R.m01 ? T(1) : T(-1)) * (t2 > T(0) ? sqrt_(t2) : T(0))};
    }
    T s, c;
    sincos_(theta, &s, &c);
    const T t = ((theta > Eps<T>::taylor) ? theta / s : T(1)) / T(2);
    return {t * (R.m21 - R.m12), t * (R.m02 - R.m20), t * (R.m10 - R.m01)};
}

// ---- spherical joints (the reference's flexibility joints): quaternions are stored x y z w like Eigen's
// quaternion::exp3 (Pinocchio v2.7.0 math/quaternion.hpp): unit quaternion of a rotation vector
template<class T> JM_DEV void quat_exp3(V3<T> v, T (&out)[4])
{
    const T t2 = dot(v, v);
    const T ts_prec = sizeof(T) == 8 ? T(1.220703125e-4) : T(1.86264515e-2);   // eps^(1/4): TaylorSeriesExpansion::precision<3>()
    T k, w;
    if (t2 > ts_prec)
    {
        const T theta = sqrt_(t2);
        T sh, ch;
        sincos_(T(0.5) * theta, &sh, &ch);
        k = sh / theta;
        w = ch;
    }
    else
    {
        k = T(0.5) - t2 * T(1.0 / 48.0);
        w = T(1) - t2 * T(0.125);
    }
    out[0] = k * v.x; out[1] = k * v.y; out[2] = k * v.z; out[3] = w;
}
// Hamilton product a * b
template<class T> JM_DEV void quat_mul(const T * a, const T (&b)[4], T (&r)[4])
{
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// quaternion::log3: rotation vector of a unit quaternion, `theta` >= 0 its angle
template<class T> JM_DEV V3<T> quat_log3(T x, T y, T z, T w, T & theta)
{
    const T n2 = x * x + y * y + z * z;
    const T n = sqrt_(n2);
    const T ts_prec = sizeof(T) == 8 ? T(1.220703125e-4) : T(1.86264515e-2);
    const T sgn = w >= T(0) ? T(1) : T(-1);
    theta = T(2) * atan2_(n, sgn * w);
    const T k = n2 > ts_prec ? sgn * theta / n : sgn * (T(2) / fabs_(w)) * (T(1) - n2 / (T(3) * w * w));
    return {k * x, k * y, k * z};
}
// Jlog3 (Pinocchio v2.7.0 spatial/explog.hpp) applied to a vector: Jlog3(theta, lg) * v
template<class T> JM_DEV V3<T> jlog3_mul(T theta, V3<T> lg, V3<T> v)
{
    const T ts_prec = sizeof(T) == 8 ? T(1.220703125e-4) : T(1.86264515e-2);
    T alpha, diag;
    if (theta < ts_prec)
    {
        alpha = T(1.0 / 12.0) + theta * theta * T(1.0 / 720.0);
        diag = T(0.5) * (T(2) - theta * theta * T(1.0 / 6.0));
    }
    else
    {
        T st, ct;
        sincos_(theta, &st, &ct);
        const T st_1mct = st / (T(1) - ct);
        alpha = T(1) / (theta * theta) - st_1mct / (T(2) * theta);
        diag = T(0.5) * (theta * st_1mct);
    }
    // (alpha lg lg^T + diag 1 + [lg/2]x) v
    return (alpha * dot(lg, v)) * lg + diag * v + T(0.5) * cross(lg, v);
}
// inverse of a symmetric positive definite 3x3 (cofactors)
template<class T> JM_DEV S3<T> sym_inverse(const S3<T> & A)
{
    const T c00 = A.yy * A.zz - A.yz * A.yz, c01 = A.xz * A.yz - A.xy * A.zz, c02 = A.xy * A.yz - A.xz * A.yy;
    const T idet = T(1) / (A.xx * c00 + A.xy * c01 + A.xz * c02);
    return {c00 * idet, c01 * idet, c02 * idet, (A.xx * A.zz - A.xz * A.xz) * idet, (A.xy * A.xz - A.xx * A.yz) * idet,
            (A.xx * A.yy - A.xy * A.xy) * idet};
}
}  // namespace jm
