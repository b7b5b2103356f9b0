// The (<= 9) x (<= 9) algebra of one DIIS step (pymes/mixer/diis.py:40-103) as plain scalar code that compiles for the
// device (one thread of a one-block kernel, kernels.hip) and for the host simulator: with it the overlaps never leave
// the GPU and the extrapolation needs no host round trip — the reference does this part with numpy on the host
// (np.linalg.eigh / inv, diis.py:85-95), which on a GPU costs a stream synchronisation and leaves the device idle while
// Python solves a 7 x 7 system.
//
// State layout (doubles): S[0] = order of the stored matrix L (number of stored vectors + 1), S[1..81] = L row-major with
// pitch 9, S[82..90] = the coefficients of the last step (c[0..m-1] amplitudes, c[m] Lagrange multiplier), S[91] = 1 if the
// last step went through the pseudo-inverse ("linear dependence found", diis.py:86), S[92] = number of steps taken.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PYMES_HD __host__ __device__
#else
#define PYMES_HD
#endif

namespace diis_small {

constexpr int kMaxOrder = 9;        // dim_space <= 8
constexpr int kStateDoubles = 96;

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix A (pitch 9): A is overwritten, lam[i] / V[:,i] on return
PYMES_HD inline void jacobi_eigh(int n, double* A, double* V, double* lam) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * 9 + j] = (i == j) ? 1.0 : 0.0;
    // Off-diagonal elements below 1e-17 of the Frobenius norm are left alone (they move an eigenvalue by less than that — the
    // callers decide at 1e-12 of an O(1) norm); a sweep without a rotation ends the iteration.  (A test on the SUM of the
    // off-diagonal squares never fires on a DIIS matrix near convergence — eigenvalues of 1e-17 next to +-2.4, the rounding
    // noise of each rotation keeps the sum above it — and all 60 sweeps ran: 40-100 us on the host with the device idle.)
    double norm2 = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) norm2 += A[i * 9 + j] * A[i * 9 + j];
    const double tiny = 1e-17 * sqrt(norm2);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotations = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * 9 + q];
                if (!(apq > tiny || apq < -tiny)) { if (apq == apq) continue; }      // (NaN rotates on: the result is NaN)
                ++rotations;
                const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / ((theta >= 0.0 ? theta : -theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {                 // A <- A J
                    const double akp = A[k * 9 + p], akq = A[k * 9 + q];
                    A[k * 9 + p] = c * akp - s * akq;
                    A[k * 9 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {                 // A <- J^T A
                    const double apk = A[p * 9 + k], aqk = A[q * 9 + k];
                    A[p * 9 + k] = c * apk - s * aqk;
                    A[q * 9 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * 9 + p], vkq = V[k * 9 + q];
                    V[k * 9 + p] = c * vkp - s * vkq;
                    V[k * 9 + q] = s * vkp + c * vkq;
                }
            }
        if (!rotations) break;
    }
    for (int i = 0; i < n; ++i) lam[i] = A[i * 9 + i];
}

// x = A^-1 b by Gaussian elimination with partial pivoting (A, b overwritten; pitch 9)
// Returns false when a pivot vanishes or is not finite (numpy.linalg.inv raises LinAlgError there): x is then not written.
PYMES_HD inline bool solve_lu(int n, double* A, double* b, double* x) {
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double big = A[k * 9 + k] < 0 ? -A[k * 9 + k] : A[k * 9 + k];
        for (int i = k + 1; i < n; ++i) {
            const double v = A[i * 9 + k] < 0 ? -A[i * 9 + k] : A[i * 9 + k];
            if (v > big) { big = v; piv = i; }
        }
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = A[k * 9 + j]; A[k * 9 + j] = A[piv * 9 + j]; A[piv * 9 + j] = t; }
            const double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        if (!(big > 0.0) || !(big <= 1.7976931348623157e308)) return false;       // zero, NaN or infinite pivot
        for (int i = k + 1; i < n; ++i) {
            const double f = A[i * 9 + k] / A[k * 9 + k];
            for (int j = k; j < n; ++j) A[i * 9 + j] -= f * A[k * 9 + j];
            b[i] -= f * b[k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= A[i * 9 + j] * x[j];
        x[i] = s / A[i * 9 + i];
    }
    return true;
}

// One DIIS step on the state S.  overlaps[t * m + i] = <e_i, e_new> of amplitude type t (i < m; the new vector is i = m-1),
// summed over the types in the order of the reference's loop (diis.py:65-78); was_full: the oldest vector has just been
// dropped (diis.py:59-60, including its quirk: the row / column of the second-newest vector is NOT carried over).
// L of this step from the stored one (S[1..81]) and the new overlaps, written back to S; returns in L (pitch 9)
PYMES_HD inline void build_L(double* S, const double* overlaps, int ntypes, int m, int was_full, double* L) {
    const double* Lold = S + 1;                   // (L is a separate array: the old matrix is read in place)
    const int n = m + 1;
    for (int i = 0; i < 81; ++i) L[i] = 0.0;
    for (int i = 0; i < m; ++i) { L[m * 9 + i] = -1.0; L[i * 9 + m] = -1.0; }                 // diis.py:56-57
    if (was_full) {                                                                            // :59-60
        for (int i = 0; i < n - 3; ++i)
            for (int j = 0; j < n - 3; ++j) L[i * 9 + j] = Lold[(i + 1) * 9 + (j + 1)];
    } else {                                                                                   // :62
        for (int i = 0; i < n - 2; ++i)
            for (int j = 0; j < n - 2; ++j) L[i * 9 + j] = Lold[i * 9 + j];
    }
    for (int i = 0; i < m; ++i) {                                                              // :65-80
        double s = 0.0;
        for (int t = 0; t < ntypes; ++t) s += overlaps[t * m + i];
        L[i * 9 + (m - 1)] += s;
    }
    for (int j = 0; j < n; ++j) L[(m - 1) * 9 + j] = L[j * 9 + (m - 1)];
    S[0] = (double)n;
    for (int i = 0; i < 81; ++i) S[1 + i] = L[i];
}

// coefficients (:82-95) from L and its eigen-decomposition (lam, V): L c = (0, ..., 0, -1); pseudo-inverse over
// |lambda| > 1e-12 when L is (nearly) singular, LU with partial pivoting otherwise (the reference inverts L there)
PYMES_HD inline void finish(double* S, const double* L, const double* V, const double* lam, int n, double* work /* [99] */) {
    double* A = work;
    double* c = work + 81;
    double* unit = work + 90;
    bool dependent = false, failed = false;
    for (int i = 0; i < n; ++i) dependent = dependent || (lam[i] < 1e-12 && lam[i] > -1e-12);
    for (int i = 0; i < n; ++i) failed = failed || !(lam[i] == lam[i]);                       // NaN overlaps
    for (int i = 0; i < n; ++i) unit[i] = (i == n - 1) ? -1.0 : 0.0;
    if (dependent) {
        for (int i = 0; i < n; ++i) c[i] = 0.0;
        for (int k = 0; k < n; ++k) {
            if (lam[k] < 1e-12 && lam[k] > -1e-12) continue;
            double proj = 0.0;
            for (int i = 0; i < n; ++i) proj += V[i * 9 + k] * unit[i];
            for (int i = 0; i < n; ++i) c[i] += V[i * 9 + k] * proj / lam[k];
        }
    } else {
        for (int i = 0; i < 81; ++i) A[i] = L[i];
        for (int i = 0; i < n; ++i) c[i] = 0.0;
        if (!solve_lu(n, A, unit, c)) failed = true;
    }
    for (int i = 0; i < n; ++i) failed = failed || !(c[i] == c[i]) || c[i] > 1.7976931348623157e308 || c[i] < -1.7976931348623157e308;
    // no finite solution: the coefficients select the newest amplitudes unchanged, so that an extrapolation that is already
    // enqueued behind this step (the device-resident form) leaves finite numbers; the host raises when it reads S[91]
    for (int i = 0; i < 9; ++i) S[82 + i] = failed ? (i == n - 2 ? 1.0 : 0.0) : (i < n ? c[i] : 0.0);
    // S[91]: 0 inverse branch, 1 pseudo-inverse branch, 2 no finite solution (a singular or non-finite L: the reference's
    // numpy.linalg.inv / eigh raise LinAlgError, diis.py:85-95) — the callers refuse to extrapolate with such coefficients
    S[91] = failed ? 2.0 : (dependent ? 1.0 : 0.0);
    S[92] += 1.0;
}

// inv = A^-1 by Gauss-Jordan elimination with partial pivoting (A overwritten; pitch 9); false when a pivot vanishes or is
// not finite
PYMES_HD inline bool invert(int n, double* A, double* inv) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) inv[i * 9 + j] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double big = A[k * 9 + k] < 0 ? -A[k * 9 + k] : A[k * 9 + k];
        for (int i = k + 1; i < n; ++i) {
            const double v = A[i * 9 + k] < 0 ? -A[i * 9 + k] : A[i * 9 + k];
            if (v > big) { big = v; piv = i; }
        }
        if (!(big > 0.0) || !(big <= 1.7976931348623157e308)) return false;
        if (piv != k)
            for (int j = 0; j < n; ++j) {
                double t = A[k * 9 + j]; A[k * 9 + j] = A[piv * 9 + j]; A[piv * 9 + j] = t;
                t = inv[k * 9 + j]; inv[k * 9 + j] = inv[piv * 9 + j]; inv[piv * 9 + j] = t;
            }
        const double d = 1.0 / A[k * 9 + k];
        for (int j = 0; j < n; ++j) { A[k * 9 + j] *= d; inv[k * 9 + j] *= d; }
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = A[i * 9 + k];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { A[i * 9 + j] -= f * A[k * 9 + j]; inv[i * 9 + j] -= f * inv[k * 9 + j]; }
        }
    }
    return true;
}

// The reference decides between inverse and pseudo-inverse by the eigenvalues of L (any |lambda| < 1e-12, diis.py:85-86).
// 1 / min|lambda| = ||L^-1||_2 <= n max|L^-1_ij|: when that bound stays below 0.5e12 no eigenvalue is within 2e-12 of zero,
// the decision is "inverse" without the eigen-decomposition (40 us of cyclic Jacobi on the host for a 7 x 7 matrix — the
// device is idle while it runs), and c = -L^-1[:, n-1].  Anything else (a large inverse, a vanished pivot, NaN) goes through
// the eigen-decomposition as before.
PYMES_HD inline void step(double* S, const double* overlaps, int ntypes, int m, int was_full) {
    double L[81], A[81], V[81], lam[9], work[99];
    const int n = m + 1;
    build_L(S, overlaps, ntypes, m, was_full, L);
    for (int i = 0; i < 81; ++i) A[i] = L[i];
    if (invert(n, A, V)) {
        double big = 0.0;
        bool finite = true;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double v = V[i * 9 + j] < 0 ? -V[i * 9 + j] : V[i * 9 + j];
                finite = finite && (v == v);
                if (v > big) big = v;
            }
        if (finite && n * big < 0.5e12) {
            for (int i = 0; i < 9; ++i) S[82 + i] = i < n ? -V[i * 9 + (n - 1)] : 0.0;
            S[91] = 0.0;
            S[92] += 1.0;
            return;
        }
    }
    for (int i = 0; i < 81; ++i) A[i] = L[i];
    jacobi_eigh(n, A, V, lam);
    finish(S, L, V, lam, n, work);
}

}  // namespace diis_small
