// Host-only check of the front end's text -> double conversion (visgeom_amd/csrc/vg_json.hpp, Cursor::number): every value must
// be the double strtod returns for the same text, bit for bit -- repr-style 17-digit numbers, pixel coordinates, 19-digit
// integers with exponents, exact half-way cases between two doubles, odd shapes.  Prints "total N bad M", exit code 1 on a mismatch.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../visgeom_amd/csrc/vg_json.hpp"

int main()
{
    std::mt19937_64 rng(12345);
    long bad = 0, total = 0;
    char buf[128];
    auto check = [&](const char *t) {
        const std::string text = t;
        vgjson::Cursor c(text.c_str(), 0, text.size());
        const double a = c.number(), b = std::strtod(t, nullptr);
        total++;
        if (std::memcmp(&a, &b, 8) != 0) {
            if (bad < 20) std::printf("MISMATCH %s -> %.17g vs %.17g\n", t, a, b);
            bad++;
        }
    };
    for (long it = 0; it < 600000; it++) {
        const int kind = (int)(it % 6);
        if (kind == 0) {
            std::snprintf(buf, sizeof buf, "%.17g", std::ldexp((double)(rng() >> 11), (int)(rng() % 80) - 60));
        } else if (kind == 1) {
            std::snprintf(buf, sizeof buf, "%.16g", (double)(rng() % 1280000) / 1000.0 + (double)(rng() % 1000000) * 1e-12);
        } else if (kind == 2) {
            std::snprintf(buf, sizeof buf, "%llue%d", (unsigned long long)(rng() % 10000000000000000000ull), (int)(rng() % 44) - 22);
        } else if (kind == 3) {  // a double plus half an ulp: the decimal expansion sits exactly between two doubles
            const double v = std::ldexp((double)((rng() >> 12) | (1ull << 52)), (int)(rng() % 20) - 60);
            const long double h = (long double)v + (long double)std::ldexp(1.0, std::ilogb(v) - 53);
            std::snprintf(buf, sizeof buf, "%.21Lg", h);
        } else if (kind == 4) {
            std::snprintf(buf, sizeof buf, "%llu", (unsigned long long)rng());
            std::string t = buf;
            const int dp = (int)(rng() % 20);
            if (dp < (int)t.size()) t.insert(t.size() - dp, ".");
            std::snprintf(buf, sizeof buf, "%s%s", (rng() & 1) ? "-" : "", t.c_str());
        } else {
            std::snprintf(buf, sizeof buf, "%.*g", (int)(rng() % 19) + 1, std::ldexp((double)(rng() >> 11), (int)(rng() % 2000) - 1000 - 53));
        }
        check(buf);
    }
    for (const char *t : {"0", "-0", "0.0", "-0.0", "1e0", "1E+2", "1e-2", "0.30000000000000004", "9007199254740993", "9007199254740992.5",
                          "18446744073709551615", "18446744073709551616", "1.7976931348623157e308", "4.9e-324", "2.2250738585072011e-308",
                          "5e-20", "12345678901234567890123", "0.000000000000000000001", "1.", "5.e3", "+3", ".5", "1e19",
                          "9999999999999999999e19", "1e-19", "0.1e-18", "1e999", "-1e999", "0x10", "1e", "1e+", "7e-400"})
        check(t);
    std::printf("total %ld bad %ld\n", total, bad);
    return bad != 0;
}
