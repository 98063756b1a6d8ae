// The model of one RANSAC trial -- estimate_fundamental_matrix (akaze/src/ops/estimate_fundamental_matrix.rs:17-69) for
// exactly eight correspondences -- as ONE piece of source for the host path (akz_ransac.cpp) and for the device kernel
// that runs the trials of match_features (akz_fmatrix.hip): f64 additions, multiplications, divisions and square roots in
// a fixed order, no contraction (-ffp-contract=off on both sides), so that the two produce the same bits.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define AKZ_HD __host__ __device__ inline
#define AKZ_UNROLL _Pragma("unroll")  // (static indices: the device keeps the 8 x 9 matrix in registers)
#else
#define AKZ_HD inline
#define AKZ_UNROLL
#endif

namespace akz {

// Thin SVD of the 8x9 design matrix A by one-sided (Hestenes) Jacobi rotations on the columns of A^T (f64; m[p] is column
// p of A^T, i.e. row p of A): the rotated columns become orthogonal, their norms are the 8 singular values and the
// normalised columns the right singular vectors of A.  Works on A itself, not on A^T A, so the condition number is not
// squared (pixel coordinates of ~1e3 give entries of ~1e6: the eigenvalue route lost the smallest singular values in
// rounding noise).
// The matrix's storage is the caller's (M: double& at(p, k)): an array on the host, LDS on the device.
struct Mat8x9 {
    double v[8][9];
    AKZ_HD double& at(int p, int k) { return v[p][k]; }
};
// one Hestenes rotation of rows p, q (columns of A^T); false: the pair is orthogonal already
template <class M>
AKZ_HD bool jacobi_pair(M& m, int p, int q) {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
    AKZ_UNROLL
    for (int k = 0; k < 9; ++k) {
        const double x = m.at(p, k), y = m.at(q, k);
        alpha += x * x;
        beta += y * y;
        gamma += x * y;
    }
    if (fabs(gamma) <= 1e-15 * sqrt(alpha * beta) || gamma == 0.0) return false;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
    AKZ_UNROLL
    for (int k = 0; k < 9; ++k) {
        const double x = m.at(p, k), y = m.at(q, k);
        m.at(p, k) = c * x - sn * y;
        m.at(q, k) = sn * x + c * y;
    }
    return true;
}
// sweeps over the pairs in row-cyclic order (0,1) (0,2) .. (0,7) (1,2) .. (6,7) until one leaves every pair alone
template <class M>
AKZ_HD void jacobi_sweeps(M& m) {
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 8; ++q) rotated = jacobi_pair(m, p, q) || rotated;
        if (!rotated) break;
    }
}
// after the sweeps: index of the smallest singular value (row norm); *full_rank = all eight exceed epsilon (as f32)
template <class M>
AKZ_HD int smallest_singular(M& m, float epsilon, bool* full_rank) {
    int rank = 0, mi = 0;
    double smallest = 0.0;
    for (int i = 0; i < 8; ++i) {
        double nrm = 0.0;
        AKZ_UNROLL
        for (int k = 0; k < 9; ++k) nrm += m.at(i, k) * m.at(i, k);
        nrm = sqrt(nrm);
        if ((float)nrm > epsilon) ++rank;
        if (i == 0 || nrm < smallest) {  // the smallest of the 8 (the first one among equals)
            smallest = nrm;
            mi = i;
        }
    }
    *full_rank = rank == 8;
    return mi;
}

// the design matrix of eight correspondences (:26-40), products in f32 as the reference forms them
template <class M>
AKZ_HD void design_matrix(M& m, const float (&x0)[8], const float (&y0)[8], const float (&x1)[8], const float (&y1)[8]) {
    for (int i = 0; i < 8; ++i) {
        const float row[9] = {x0[i] * x1[i], x0[i] * y1[i], x0[i], y0[i] * x1[i], y0[i] * y1[i], y0[i], x1[i], y1[i], 1.0f};
        AKZ_UNROLL
        for (int k = 0; k < 9; ++k) m.at(i, k) = (double)row[k];
    }
}
// the model from the rotated matrix: the right singular vector of the smallest of the 8 singular values, as
// [[v0, v3, v6], [v1, v4, v7], [v2, v5, v8]] (:46-66).  false: rank < 8 at `epsilon` (the reference's None, :44).
template <class M>
AKZ_HD bool model_from_rotated(M& m, float epsilon, float (&f)[9]) {
    bool full = false;
    const int mi = smallest_singular(m, epsilon, &full);
    if (!full) return false;
    double nrm = 0.0;
    AKZ_UNROLL
    for (int k = 0; k < 9; ++k) nrm += m.at(mi, k) * m.at(mi, k);
    nrm = sqrt(nrm);
    float v[9];
    AKZ_UNROLL
    for (int k = 0; k < 9; ++k) v[k] = (float)(nrm > 0.0 ? m.at(mi, k) / nrm : 0.0);
    f[0] = v[0]; f[1] = v[3]; f[2] = v[6];
    f[3] = v[1]; f[4] = v[4]; f[5] = v[7];
    f[6] = v[2]; f[7] = v[5]; f[8] = v[8];
    return true;
}

// x0, y0 from keypoints_0 and x1, y1 from keypoints_1 of the eight sampled matches -> the model, row-major
AKZ_HD bool fundamental_from_8(const float (&x0)[8], const float (&y0)[8], const float (&x1)[8], const float (&y1)[8], float epsilon,
                               float (&f)[9]) {
    Mat8x9 m;
    design_matrix(m, x0, y0, x1, y1);
    jacobi_sweeps(m);
    return model_from_rotated(m, epsilon, f);
}

// evaluate_model (:79-83): |p_r^T F p_l| with p_l = (x0, y0, 1), p_r = (x1, y1, 1); the products with 1 are exact
AKZ_HD float fundamental_error(const float (&f)[9], float x0, float y0, float x1, float y1) {
    const float r0 = (x1 * f[0] + y1 * f[3]) + f[6], r1 = (x1 * f[1] + y1 * f[4]) + f[7], r2 = (x1 * f[2] + y1 * f[5]) + f[8];
    const float s = (r0 * x0 + r1 * y0) + r2;
    return fabsf(s);
}

}  // namespace akz
