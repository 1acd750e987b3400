// format_g6_check.cpp -- vgtext::fmt_g6 (visgeom_amd/csrc/vg_text_format.hpp) against snprintf("%g") -- test infrastructure.
// Prints "checked N bad B"; exit code 1 when any text differs.  Built and run by tests/test_frontend_cpu.py.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../visgeom_amd/csrc/vg_text_format.hpp"

static long long checked = 0, bad = 0;

static void check(double v)
{
    char a[64], b[64];
    const int la = vgtext::fmt_g6(v, a, sizeof a);
    const int lb = std::snprintf(b, sizeof b, "%g", v);
    checked++;
    if (la != lb || std::memcmp(a, b, (size_t)la) != 0) {
        if (bad < 20) std::printf("MISMATCH %.17g: got '%.*s' want '%s'\n", v, la, a, b);
        bad++;
    }
}

static void around(double v)
{
    double lo = v, hi = v;
    check(v);
    check(-v);
    for (int k = 0; k < 3; k++) {
        lo = std::nextafter(lo, -std::numeric_limits<double>::infinity());
        hi = std::nextafter(hi, std::numeric_limits<double>::infinity());
        check(lo);
        check(hi);
    }
}

int main()
{
    std::mt19937_64 g(20260929);
    std::uniform_real_distribution<double> U(0., 1.);
    std::normal_distribution<double> Nrm(0., 0.1);
    // specials
    for (double v : {0., -0., 1., -1., 0.5, 1e-5, 1e-4, 9.9999949e-5, 9.9999951e-5, 0.0001, 100000., 999999., 999999.4, 999999.5, 999999.6,
                     1e6, 1e15, 1e16, 1e-300, 5e-324, 1.7976931348623157e308, 123456.5, 123457.5, 0.1234565, 2.5, 1e22, 1e23})
        around(v);
    check(std::numeric_limits<double>::infinity());
    check(-std::numeric_limits<double>::infinity());
    // powers of ten and the last six-digit number below them, with neighbours
    for (int e = -30; e <= 30; e++) {
        around(std::pow(10., e));
        around(9.999995 * std::pow(10., e));
        around(9.9999949999 * std::pow(10., e));
        around(1.000005 * std::pow(10., e));
    }
    // ties and near-ties of the sixth digit: (D + 0.5) 10^k, exact when representable
    for (int i = 0; i < 400000; i++) {
        const double D = 100000. + std::floor(U(g) * 900000.);
        const int k = (int)std::floor(U(g) * 24.) - 14;
        around((D + 0.5) * std::pow(10., k));
        around((D + 0.5) / std::pow(10., -k));
    }
    // what the residual report prints: residuals of ~0.1 px, pixel coordinates, metres, radians
    for (int i = 0; i < 3000000; i++) {
        check(Nrm(g));
        check(U(g) * 1280.);
        check((U(g) - 0.5) * 3.);
        check(std::floor(U(g) * 2000.) / 8.);   // short decimals
    }
    // log-uniform magnitudes, both signs
    for (int i = 0; i < 3000000; i++) {
        const double v = std::pow(10., U(g) * 60. - 30.) * (U(g) < 0.5 ? -1. : 1.);
        check(v);
    }
    // integers
    for (int i = 0; i < 300000; i++) check(std::floor(U(g) * 3e6) - 1e6);

    // speed (informative)
    std::vector<double> xs(1 << 20);
    for (double &x : xs) x = U(g) * 1280.;
    char buf[64];
    long long sink = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (double x : xs) sink += vgtext::fmt_g6(x, buf, sizeof buf);
    auto t1 = std::chrono::steady_clock::now();
    for (double x : xs) sink += vgtext::fmt_g6_exact(x, buf, sizeof buf);
    auto t2 = std::chrono::steady_clock::now();
    std::printf("fmt_g6 %.1f ns, to_chars %.1f ns per number (%lld)\n", std::chrono::duration<double>(t1 - t0).count() * 1e9 / xs.size(),
                std::chrono::duration<double>(t2 - t1).count() * 1e9 / xs.size(), sink);
    std::printf("checked %lld bad %lld\n", checked, bad);
    return bad ? 1 : 0;
}
