// TEST INFRASTRUCTURE (CPU oracle). Bound-constrained trust-region solver for the trim problem.
//
// The reference minimises cost(z) = sum r_i(z)^2 over the 7 trim unknowns with NLopt's :LN_BOBYQA inside box
// bounds, initial_step = 0.05, stopval = 1e-16, maxeval = 100000 (FlightApps/src/c172/c172.jl:883-942). BOBYQA
// (NLopt 2.x, a third-party dependency absent from /root/reference) is a derivative-free trust-region method: it
// starts with a trust radius equal to the initial step, never leaves the box, moves at most one radius per
// iteration and therefore follows a continuous descent path from TrimState() to the nearest zero of the cost.
// Its iterates are not restated here; what is restated is that behaviour, for a zero-residual problem:
//   * a box-shaped trust region |dz|_inf <= D, D0 = 0.05 (the reference's initial_step), intersected with the bounds;
//   * inside it the Gauss-Newton model |r + J dz|^2 is minimised subject to the box (active-set bounded least
//     squares: a variable that would leave the box is held at its face and released again when the model's
//     gradient points back inside), so the iterate slides along a bound instead of being clamped onto it;
//   * D doubles after a step the model predicted well, shrinks after a rejected one;
//   * the engine and aerodynamic maps are piecewise linear, so r is only piecewise smooth: where the forward and the
//     backward difference of a column disagree (a knot within one difference step of the iterate; n_eng = 1.074 is
//     such a knot, piston.jl:110,140) the column is replaced by the pure one-sided slopes taken one step further out,
//     and the step is computed with the side it moves to (a semismooth Newton step) instead of the central average,
//     which straddles the kink and makes the iteration creep; the mirror choice is tried too and the better kept;
//   * success <=> cost <= 1e-16 (the reference's STOPVAL_REACHED test); iteration continues to the rounding floor
//     so that independent implementations land on the same zero to ~1e-12.
// If the descent from TrimState() stalls above stopval (a kink of the piecewise-linear engine maps can trap a
// local model), the solve is repeated by continuation in the trim PARAMETERS from the reference's default
// TrimParameters() — whose trim from TrimState() is pinned by FAt/c172/test_c172s.jl:22-38 — towards the requested
// ones: the same solver warm-started along the segment. A point reachable that way has a trim, and BOBYQA's
// maxeval = 1e5 budget (~7000 trust-region iterations) finds it; a point where even continuation stops short is
// reported as failed, like the reference's `success = false`.
#pragma once
#include <algorithm>
#include <cmath>

namespace fo {

struct TrimSolveStats { int iters = 0; int evals = 0; bool continued = false; };

// Minimises |r(z)|^2, r: R^7 -> R^7, inside [lo, hi]. Returns the final cost; z is updated in place.
namespace trimdetail {
constexpr int N = 7;
// min over dl <= d <= du of d'Hd + 2 g'd (H = J'J, g = J'r) by an active set: a variable that would leave the box is held
// at the face it hits, and released when the model gradient there points back inside
inline void box_gauss_newton(const double (&H)[N][N], const double (&g)[N], const double* dl, const double* du, double ridge, double* d) {
    int fixed[N];
    for (int k = 0; k < N; k++) { d[k] = 0; fixed[k] = 0; }
    for (int pass = 0; pass < 4 * N; pass++) {
        int idx[N], nf = 0;
        for (int k = 0; k < N; k++) if (!fixed[k]) idx[nf++] = k;
        double A[N][N + 1];
        for (int a = 0; a < nf; a++) {   // H_FF d_F = -(g_F + H_FB d_B)
            double rhs = -g[idx[a]];
            for (int k = 0; k < N; k++) if (fixed[k]) rhs -= H[idx[a]][k] * d[k];
            for (int b = 0; b < nf; b++) A[a][b] = H[idx[a]][idx[b]] + (a == b ? ridge : 0.0);
            A[a][nf] = rhs;
        }
        for (int c = 0; c < nf; c++) {   // Gaussian elimination, partial pivoting
            int p = c;
            for (int q = c + 1; q < nf; q++) if (std::fabs(A[q][c]) > std::fabs(A[p][c])) p = q;
            if (p != c) for (int q = 0; q <= nf; q++) std::swap(A[p][q], A[c][q]);
            const double piv = A[c][c] != 0 ? A[c][c] : 1e-300;
            for (int q = c + 1; q < nf; q++) {
                const double f = A[q][c] / piv;
                for (int w = c; w <= nf; w++) A[q][w] -= f * A[c][w];
            }
        }
        double sol[N];
        for (int q = nf - 1; q >= 0; q--) {
            double s = A[q][nf];
            for (int w = q + 1; w < nf; w++) s -= A[q][w] * sol[w];
            sol[q] = s / (A[q][q] != 0 ? A[q][q] : 1e-300);
        }
        double t = 1.0;   // longest feasible fraction of the move towards the free minimum
        int hit = -1, side = 0;
        for (int a = 0; a < nf; a++) {
            const int k = idx[a];
            const double delta = sol[a] - d[k];
            if (delta > 0 && d[k] + delta > du[k]) { const double tt = (du[k] - d[k]) / delta; if (tt < t) { t = tt; hit = k; side = 1; } }
            if (delta < 0 && d[k] + delta < dl[k]) { const double tt = (dl[k] - d[k]) / delta; if (tt < t) { t = tt; hit = k; side = -1; } }
        }
        for (int a = 0; a < nf; a++) { const int k = idx[a]; d[k] += t * (sol[a] - d[k]); }
        if (hit >= 0) { fixed[hit] = side; d[hit] = side > 0 ? du[hit] : dl[hit]; continue; }
        int rel = -1;
        double best = 0;
        for (int k = 0; k < N; k++) if (fixed[k]) {
            double gm = g[k];
            for (int b = 0; b < N; b++) gm += H[k][b] * d[b];
            const double inward = fixed[k] > 0 ? gm : -gm;   // at the upper face a positive gradient wants to come back
            if (inward > best) { best = inward; rel = k; }
        }
        if (rel < 0 || best <= 1e-14 * (std::fabs(g[rel]) + 1e-300)) break;
        fixed[rel] = 0;
    }
}
}  // namespace trimdetail

template <class Resid>
inline double trim_tr_minimize(Resid&& resid, const double* lo, const double* hi, double* z, int max_iter, TrimSolveStats* stats = nullptr) {
    using trimdetail::N;
    const double fd = 1e-6;          // difference step
    const double cost_floor = 1e-27; // well below stopval: independent implementations converge onto the same zero
    double r[N];
    for (int k = 0; k < N; k++) z[k] = std::clamp(z[k], lo[k], hi[k]);
    resid(z, r);
    double cost = 0;
    for (int k = 0; k < N; k++) cost += r[k] * r[k];
    double D = 0.05;                 // the reference's initial_step (c172.jl:919)
    int it = 0, evals = 1;
    for (; it < max_iter && cost > cost_floor && D > 1e-13; it++) {
        // central difference columns (one-sided at a bound: the inputs are Ranged, nothing exists outside); where the forward
        // and backward differences disagree, Jf / Jb hold the slopes over [z+h, z+2h] / [z-2h, z-h]
        double Jc[N][N], Jf[N][N], Jb[N][N];
        bool kink[N], any_kink = false;
        for (int j = 0; j < N; j++) {
            double zz[N], rp[N], rm[N];
            for (int k = 0; k < N; k++) zz[k] = z[k];
            const double zp = std::min(z[j] + fd, hi[j]), zm = std::max(z[j] - fd, lo[j]);
            zz[j] = zp; resid(zz, rp);
            zz[j] = zm; resid(zz, rm);
            evals += 2;
            const double ic = 1.0 / (zp - zm);
            const double ifw = zp > z[j] ? 1.0 / (zp - z[j]) : 0.0, ibw = z[j] > zm ? 1.0 / (z[j] - zm) : 0.0;
            double dmax = 0, cmax = 0;
            for (int i = 0; i < N; i++) {
                Jc[i][j] = (rp[i] - rm[i]) * ic;
                Jf[i][j] = ifw != 0 ? (rp[i] - r[i]) * ifw : Jc[i][j];
                Jb[i][j] = ibw != 0 ? (r[i] - rm[i]) * ibw : Jc[i][j];
                dmax = std::max(dmax, std::fabs(Jf[i][j] - Jb[i][j]));
                cmax = std::max(cmax, std::fabs(Jc[i][j]));
            }
            kink[j] = dmax > 1e-3 * cmax;   // smooth: |Jf - Jb| ~ fd |r''| ~ 1e-6 of the column
            if (kink[j]) {
                any_kink = true;
                double r2[N];
                if (zp + fd <= hi[j]) { zz[j] = zp + fd; resid(zz, r2); evals++; for (int i = 0; i < N; i++) Jf[i][j] = (r2[i] - rp[i]) / (zp + fd - zp); }
                if (zm - fd >= lo[j]) { zz[j] = zm - fd; resid(zz, r2); evals++; for (int i = 0; i < N; i++) Jb[i][j] = (rm[i] - r2[i]) / (zm - (zm - fd)); }
            }
        }
        bool accepted = false;
        for (int attempt = 0; attempt < 40 && !accepted && D > 1e-13; attempt++) {
            double dl[N], du[N];
            for (int k = 0; k < N; k++) { dl[k] = std::max(lo[k] - z[k], -D); du[k] = std::min(hi[k] - z[k], D); }
            // candidate 0: sides made consistent with the step, starting from the central columns; candidate 1: the mirror sides
            double best_cn = 0, best_pred = 0, best_dinf = 0, best_zn[N], best_rn[N];
            bool have = false;
            int side0[N];
            for (int cand = 0; cand < (any_kink ? 2 : 1); cand++) {
                double J[N][N], H[N][N], g[N], d[N];
                int side[N];
                for (int j = 0; j < N; j++) {
                    side[j] = cand == 0 ? 0 : (kink[j] ? (side0[j] > 0 ? -1 : 1) : 0);
                    for (int i = 0; i < N; i++) J[i][j] = side[j] == 0 ? Jc[i][j] : side[j] > 0 ? Jf[i][j] : Jb[i][j];
                }
                for (int round = 0; round < (cand == 0 ? 3 : 1); round++) {
                    double tr = 0;
                    for (int a = 0; a < N; a++) {
                        g[a] = 0;
                        for (int i = 0; i < N; i++) g[a] += J[i][a] * r[i];
                        for (int b = 0; b < N; b++) {
                            double sum = 0;
                            for (int i = 0; i < N; i++) sum += J[i][a] * J[i][b];
                            H[a][b] = sum;
                        }
                        tr += H[a][a];
                    }
                    trimdetail::box_gauss_newton(H, g, dl, du, 1e-14 * tr + 1e-300, d);
                    if (cand != 0) break;
                    bool changed = false;
                    for (int j = 0; j < N; j++) if (kink[j]) {
                        const int want = d[j] > 0 ? 1 : d[j] < 0 ? -1 : (side[j] != 0 ? side[j] : 1);
                        if (want != side[j]) {
                            side[j] = want; changed = true;
                            for (int i = 0; i < N; i++) J[i][j] = want > 0 ? Jf[i][j] : Jb[i][j];
                        }
                    }
                    if (!changed) break;
                }
                if (cand == 0) for (int j = 0; j < N; j++) side0[j] = side[j];
                double pred = 0, dinf = 0;   // predicted reduction of the model actually used
                for (int a = 0; a < N; a++) {
                    double Hd = 0;
                    for (int b = 0; b < N; b++) Hd += H[a][b] * d[b];
                    pred -= d[a] * (2 * g[a] + Hd);
                    dinf = std::max(dinf, std::fabs(d[a]));
                }
                if (!(pred > 0) || dinf == 0) continue;
                double zn[N], rn[N];
                for (int k = 0; k < N; k++) zn[k] = std::clamp(z[k] + d[k], lo[k], hi[k]);
                resid(zn, rn);
                evals++;
                double cn = 0;
                for (int k = 0; k < N; k++) cn += rn[k] * rn[k];
                if (!have || cn < best_cn) {
                    have = true; best_cn = cn; best_pred = pred; best_dinf = dinf;
                    for (int k = 0; k < N; k++) { best_zn[k] = zn[k]; best_rn[k] = rn[k]; }
                }
            }
            if (!have) { D *= 0.25; continue; }
            const double rho = (cost - best_cn) / best_pred;
            if (best_cn < cost) {
                for (int k = 0; k < N; k++) { z[k] = best_zn[k]; r[k] = best_rn[k]; }
                cost = best_cn;
                accepted = true;
                if (rho > 0.75 && best_dinf > 0.9 * D) D = std::min(2 * D, 1.0);
                else if (rho < 0.25) D = std::max(0.5 * best_dinf, 1e-14);
            } else {
                D = 0.25 * std::min(D, best_dinf);
            }
        }
        if (!accepted) break;
    }
    if (stats) { stats->iters += it; stats->evals += evals; }
    return cost;
}

}  // namespace fo
