// se3_exp_series and se3_exp (rumi_slam_amd/csrc/opt_math.h, host compile) against the closed form evaluated in long double (sinl / cosl: good
// to ~1e-19 here): the largest error of the quaternion and translation components over random twists with |omega| in [1e-5, 0.5], and the
// hand-over to the general form outside that range.  (The matrix form's c = (theta - sin theta) / theta^3 cancels at small theta -- g2o's
// formula, which the oracle shares -- so ITS translation is only good to ~1e-11 there; the series form is the accurate one.)
//   g++ -O2 -std=c++17 -I rumi_slam_amd/csrc tests/cpp/opt_math_check.cc -o /tmp/opt_math_check && /tmp/opt_math_check
#include <cstdio>
#include <cstdlib>
#include <random>

#include "opt_math.h"

int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0), L(-5.0, -0.30103);      // log10 |omega|: 1e-5 .. 0.5
    double worst = 0, worstMatrix = 0;
    for (int it = 0; it < 200000; it++) {
        double u[6];
        double d[3] = {U(rng), U(rng), U(rng)};
        const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (n < 1e-3) continue;
        const double th = std::pow(10.0, L(rng));
        for (int k = 0; k < 3; k++) u[k] = d[k] / n * th;
        for (int k = 3; k < 6; k++) u[k] = 3.0 * U(rng);
        const rumi::DSE3 a = rumi::se3_exp(u), b = rumi::se3_exp_series(u);
        // closed form in long double
        const long double wx = u[0], wy = u[1], wz = u[2], t2 = wx * wx + wy * wy + wz * wz, t = sqrtl(t2);
        const long double kq = sinl(t / 2) / t, cb = (1 - cosl(t)) / t2, cc = (t - sinl(t)) / (t2 * t);
        const long double ux = u[3], uy = u[4], uz = u[5];
        const long double c1x = wy * uz - wz * uy, c1y = wz * ux - wx * uz, c1z = wx * uy - wy * ux;
        const long double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
        const long double tr[7] = {kq * wx, kq * wy, kq * wz, cosl(t / 2), ux + cb * c1x + cc * c2x, uy + cb * c1y + cc * c2y, uz + cb * c1z + cc * c2z};
        const double va[7] = {a.r.x, a.r.y, a.r.z, a.r.w, a.t.x, a.t.y, a.t.z}, vb[7] = {b.r.x, b.r.y, b.r.z, b.r.w, b.t.x, b.t.y, b.t.z};
        for (int k = 0; k < 7; k++) {
            worst = std::fmax(worst, (double)fabsl(vb[k] - tr[k]) / (k < 4 ? 1.0 : 3.0));
            worstMatrix = std::fmax(worstMatrix, (double)fabsl(va[k] - tr[k]) / (k < 4 ? 1.0 : 3.0));
        }
    }
    int handover = 1;
    for (double th : {0.0, 1e-9, 9.9e-6, 0.51, 1.5, 3.0}) {                      // outside the range: the general form itself
        const double u[6] = {th * 0.6, -th * 0.48, th * 0.64, 0.3, -0.2, 0.9};
        const rumi::DSE3 a = rumi::se3_exp(u), b = rumi::se3_exp_series(u);
        if (a.r.x != b.r.x || a.r.y != b.r.y || a.r.z != b.r.z || a.r.w != b.r.w || a.t.x != b.t.x || a.t.y != b.t.y || a.t.z != b.t.z) handover = 0;
    }
    std::printf("series form: worst error %.3e; matrix form: %.3e; handover %d\n", worst, worstMatrix, handover);
    return worst < 1e-14 && handover ? 0 : 1;
}
