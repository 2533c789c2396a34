// tr2d.hpp -- host-side scalar pieces of the trust-region-reflective outer loop.
//
// Restates the small dense helpers of SCIPY/optimize/_lsq/common.py that trf_no_bounds
// (SCIPY/optimize/_lsq/trf.py:401-560) calls once per outer iteration; everything that touches an
// n- or m-vector runs on the GPU, these only see 2x2 systems and scalars.
#pragma once

#include <cmath>
#include <algorithm>

#if defined(__HIPCC__)
#define SFMBA_HD __host__ __device__
#else
#define SFMBA_HD
#endif

namespace sfmba {

// min 0.5 p^T B p + g^T p  subject to |p| <= Delta, B symmetric 2x2 given as (b11, b12, b22).
// Same minimiser as solve_trust_region_2d (common.py:171-219): the Newton step when B is positive
// definite and the step is inside the region, otherwise the best point on the boundary.  scipy
// finds the latter from the real roots of a quartic (numpy.roots); here it comes from the secular
// equation in the eigenbasis of B, which has the same global solution.
// Returns true when the Newton step was taken.
SFMBA_HD inline bool solve_trust_region_2d(const double B[3], const double g[2], double Delta, double p[2]) {
    const double b11 = B[0], b12 = B[1], b22 = B[2];
    const double det = b11 * b22 - b12 * b12;
    if (b11 > 0.0 && det > 0.0) {
        // Cholesky succeeds exactly when b11 > 0 and the Schur complement b22 - b12^2/b11 > 0
        const double p0 = -(b22 * g[0] - b12 * g[1]) / det;
        const double p1 = -(-b12 * g[0] + b11 * g[1]) / det;
        if (p0 * p0 + p1 * p1 <= Delta * Delta) { p[0] = p0; p[1] = p1; return true; }
    }
    if (!(Delta > 0.0)) { p[0] = p[1] = 0.0; return false; }
    // eigen-decomposition B = Q diag(l1, l2) Q^T with l1 <= l2
    const double half_tr = 0.5 * (b11 + b22);
    const double rad = sqrt(0.25 * (b11 - b22) * (b11 - b22) + b12 * b12);
    const double l1 = half_tr - rad, l2 = half_tr + rad;
    double q1x, q1y;                        // eigenvector of l1
    if (fabs(b12) > 0.0) {
        // (B - l1 I) q = 0  ->  q ~ (b12, l1 - b11) or (l1 - b22, b12); take the better conditioned
        const double ax = b12, ay = l1 - b11, bx = l1 - b22, by = b12;
        if (ax * ax + ay * ay >= bx * bx + by * by) { q1x = ax; q1y = ay; } else { q1x = bx; q1y = by; }
        const double nrm = sqrt(q1x * q1x + q1y * q1y);
        q1x /= nrm; q1y /= nrm;
    } else if (b11 <= b22) { q1x = 1.0; q1y = 0.0; } else { q1x = 0.0; q1y = 1.0; }
    const double q2x = -q1y, q2y = q1x;     // eigenvector of l2
    const double h1 = q1x * g[0] + q1y * g[1];
    const double h2 = q2x * g[0] + q2y * g[1];
    const double gn = sqrt(h1 * h1 + h2 * h2);
    if (gn == 0.0) {                        // pure quadratic: move along the smallest eigenvector
        if (l1 < 0.0) { p[0] = Delta * q1x; p[1] = Delta * q1y; } else { p[0] = p[1] = 0.0; }
        return false;
    }
    // secular equation: phi(lam) = h1^2/(l1+lam)^2 + h2^2/(l2+lam)^2 - Delta^2 = 0, lam >= max(0,-l1)
    const double lam_min = fmax(0.0, -l1);
    auto norm2 = [&](double lam) {
        const double d1 = l1 + lam, d2 = l2 + lam;
        return h1 * h1 / (d1 * d1) + h2 * h2 / (d2 * d2);
    };
    // hard case: no component along q1 and the constrained step at lam_min is still inside
    const double tiny = 1e-300;
    if (fabs(h1) <= 1e-16 * gn) {
        const double d2 = l2 + lam_min;
        if (d2 > tiny) {
            const double c2 = -h2 / d2;
            if (c2 * c2 <= Delta * Delta) {
                const double tau = sqrt(fmax(0.0, Delta * Delta - c2 * c2));
                p[0] = c2 * q2x + tau * q1x;
                p[1] = c2 * q2y + tau * q1y;
                return false;
            }
        }
    }
    double lo = lam_min, hi = gn / Delta - l1;       // |p(lam)| <= gn / (l1 + lam)
    if (hi < lo) hi = lo;
    double lam = hi;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        // Newton on 1/|p| (More-Sorensen), safeguarded by the bracket
        const double d1 = l1 + lam, d2 = l2 + lam;
        double next = mid;
        if (d1 > tiny && d2 > tiny) {
            const double n2 = h1 * h1 / (d1 * d1) + h2 * h2 / (d2 * d2);
            const double dn2 = -2.0 * (h1 * h1 / (d1 * d1 * d1) + h2 * h2 / (d2 * d2 * d2));
            const double n = sqrt(n2);
            // f = 1/Delta - 1/n ; f' = 0.5 dn2 / n^3
            const double f = 1.0 / Delta - 1.0 / n;
            const double fp = 0.5 * dn2 / (n2 * n);
            if (fp != 0.0) next = lam - f / fp;
        }
        if (!(next > lo && next < hi)) next = mid;
        if (norm2(next) > Delta * Delta) lo = next; else hi = next;
        const double prev = lam;
        lam = next;
        if (fabs(lam - prev) <= 1e-16 * fmax(1.0, fabs(lam)) || hi - lo <= 1e-16 * hi) break;
    }
    const double d1 = l1 + lam, d2 = l2 + lam;
    const double c1 = d1 > tiny ? -h1 / d1 : 0.0;
    const double c2 = d2 > tiny ? -h2 / d2 : 0.0;
    double px = c1 * q1x + c2 * q2x, py = c1 * q1y + c2 * q2y;
    const double nrm = sqrt(px * px + py * py);
    if (nrm > 0.0) { px *= Delta / nrm; py *= Delta / nrm; }   // land exactly on the boundary
    p[0] = px; p[1] = py;
    return false;
}

// update_tr_radius, common.py:222-245
SFMBA_HD inline double update_tr_radius(double Delta, double actual, double predicted, double step_norm,
                               bool bound_hit, double* ratio_out) {
    double ratio;
    if (predicted > 0.0) ratio = actual / predicted;
    else if (predicted == 0.0 && actual == 0.0) ratio = 1.0;
    else ratio = 0.0;
    if (ratio < 0.25) Delta = 0.25 * step_norm;
    else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
    *ratio_out = ratio;
    return Delta;
}

// check_termination, common.py:705-717 (0 = continue)
SFMBA_HD inline int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio,
                             double ftol, double xtol) {
    const bool f_ok = dF < ftol * F && ratio > 0.25;
    const bool x_ok = dx_norm < xtol * (xtol + x_norm);
    if (f_ok && x_ok) return 4;
    if (f_ok) return 2;
    if (x_ok) return 3;
    return 0;
}

// ---- the 2-D subspace model of trf.py:481-485 from dot products, and one trust-region step in it -----
// Inputs (all rank-reduced scalars): a11 = |g_h|^2, a12 = g_h.gn_h, a22 = |gn_h|^2 (scaled space),
// G11 = |J_h g_h|^2, G12 = (J_h g_h).(J_h gn_h), G22 = |J_h gn_h|^2, and for the norm of the unscaled step
// b11 = |D^2 g|^2, b12 = (D^2 g).p, b22 = |p|^2.  span(g_h, gn_h) is orthonormalised by Gram-Schmidt.
struct TrModel {
    double B[3], gS[2];
    double s11, r12, r22, b11, b12, b22;
    int two_d;
};

SFMBA_HD inline TrModel tr_build_model(double G11, double G12, double G22, double a11, double a12, double a22,
                                       double b11, double b12, double b22) {
    TrModel m;
    m.s11 = sqrt(a11);
    m.r12 = a12 / m.s11;
    const double r22sq = a22 - m.r12 * m.r12;
    m.two_d = (r22sq > 1e-28 * a22 && r22sq > 0.0) ? 1 : 0;
    m.r22 = m.two_d ? sqrt(r22sq) : 1.0;
    m.gS[0] = m.s11; m.gS[1] = 0.0;
    m.B[0] = G11 / a11;
    if (m.two_d) {
        m.B[1] = (G12 / m.s11 - m.r12 * G11 / a11) / m.r22;
        m.B[2] = (G22 - 2.0 * m.r12 * G12 / m.s11 + m.r12 * m.r12 * G11 / a11) / (m.r22 * m.r22);
    } else {                                    // gn_h parallel to g_h (or zero): 1-D model along g_h
        m.B[1] = 0.0; m.B[2] = 1.0;
    }
    m.b11 = b11; m.b12 = b12; m.b22 = b22;
    return m;
}

// step_h = c1 g_h + c2 gn_h, step = D step_h = c1 D^2 g + c2 p   (trf.py:493-497)
struct TrStep { double c1, c2, predicted, step_h_norm, step_norm; };

SFMBA_HD inline TrStep tr_solve_step(const TrModel& m, double Delta) {
    double pS[2];
    solve_trust_region_2d(m.B, m.gS, Delta, pS);
    if (!m.two_d) pS[1] = 0.0;
    TrStep s;
    s.predicted = -(0.5 * (m.B[0] * pS[0] * pS[0] + 2.0 * m.B[1] * pS[0] * pS[1] + m.B[2] * pS[1] * pS[1]) +
                    m.gS[0] * pS[0] + m.gS[1] * pS[1]);
    s.c2 = m.two_d ? pS[1] / m.r22 : 0.0;
    s.c1 = (pS[0] - (m.two_d ? pS[1] * m.r12 / m.r22 : 0.0)) / m.s11;
    s.step_h_norm = sqrt(pS[0] * pS[0] + pS[1] * pS[1]);
    s.step_norm = sqrt(fmax(0.0, s.c1 * s.c1 * m.b11 + 2.0 * s.c1 * s.c2 * m.b12 + s.c2 * s.c2 * m.b22));
    return s;
}

}  // namespace sfmba
