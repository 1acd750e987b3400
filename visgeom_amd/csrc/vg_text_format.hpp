// vg_text_format.hpp -- numbers -> text for the front end's reports (host only, no device code).
//
// The reference prints through operator<<(ostream, double) with the default precision: the text printf's "%g" produces
// (writeImageResidual unified_calibration.cpp:1210-1213, operator<<(Transformation) transformation.h:141-145).  At 10 000 images
// the residual report is 960 000 lines of ten numbers, and the conversion was half of the product's wall clock
// (profiles/NOTES.md round 5): std::to_chars(general, 6) is exact and costs ~300 ns per number on the GPU boxes' hosts.
// fmt_g6() below produces the same characters: six significant digits from ONE correctly rounded multiplication (or division)
// by an exact power of ten, accepted only when the scaled value is provably not near a rounding tie; everything else (ties,
// zero, subnormal / huge magnitudes, inf, nan) takes to_chars.  tests/host/format_g6_check.cpp holds it to snprintf("%g") on
// millions of values including exact ties and decade boundaries.
#pragma once

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace vgtext {

// the exact route: "%g" by definition
inline int fmt_g6_exact(double v, char *buf, size_t size)
{
    const std::to_chars_result r = std::to_chars(buf, buf + size, v, std::chars_format::general, 6);
    if (r.ec != std::errc()) return std::snprintf(buf, size, "%g", v);
    return (int)(r.ptr - buf);
}

// v in "%g" form into buf (at least 32 bytes), returns the length.  No terminating zero.
inline int fmt_g6(double v, char *buf, size_t size = 32)
{
    static const double p10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};   // all exact doubles
    const double a = std::fabs(v);
    if (!(a >= 1e-5 && a < 1e15) || size < 24) return fmt_g6_exact(v, buf, size);
    uint64_t bits;
    std::memcpy(&bits, &a, sizeof bits);
    const int e2 = (int)(bits >> 52) - 1023;                 // floor(log2 a)
    int X = (int)std::floor(e2 * 0.30102999566398120);       // floor(log10 2^e2): floor(log10 a) or one less
    // y = a * 10^(5 - X): ONE rounding (10^|s| is exact for |s| <= 22), |y - exact| <= y 2^-53 < 1.2e-10
    auto scaled = [&](int x) {
        const int s = 5 - x;
        return s >= 0 ? a * p10[s] : a / p10[-s];
    };
    double y = scaled(X);
    if (y >= 1e6) y = scaled(++X);
    else if (y < 1e5) y = scaled(--X);
    if (!(y >= 1e5 && y < 1e6)) return fmt_g6_exact(v, buf, size);
    const double fl = std::floor(y), fr = y - fl;            // exact
    if (std::fabs(fr - 0.5) < 1e-6) return fmt_g6_exact(v, buf, size);   // a tie or too close to one for y's error: exact route
    uint32_t D = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);        // the six significant digits, correctly rounded
    if (D == 1000000u) {                                     // 999999.6 -> 1.00000e(X+1)
        D = 100000u;
        X++;
    }
    char d[6];
    for (int i = 5; i >= 0; i--) {
        d[i] = (char)('0' + D % 10u);
        D /= 10u;
    }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') nd--;                 // "%g" removes trailing zeros of the fraction
    char *o = buf;
    if (std::signbit(v)) *o++ = '-';
    if (X < -4 || X >= 6) {                                  // d[.ddddd]e+XX
        *o++ = d[0];
        if (nd > 1) {
            *o++ = '.';
            for (int i = 1; i < nd; i++) *o++ = d[i];
        }
        *o++ = 'e';
        int ax = X;
        if (ax < 0) {
            *o++ = '-';
            ax = -ax;
        } else {
            *o++ = '+';
        }
        *o++ = (char)('0' + ax / 10);                        // |X| <= 14 here: two digits, as printf pads
        *o++ = (char)('0' + ax % 10);
    } else if (X >= 0) {                                     // X + 1 integer digits, the rest behind the point
        for (int i = 0; i <= X; i++) *o++ = d[i];
        if (nd > X + 1) {
            *o++ = '.';
            for (int i = X + 1; i < nd; i++) *o++ = d[i];
        }
    } else {                                                 // 0.000ddd
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -X - 1; i++) *o++ = '0';
        for (int i = 0; i < nd; i++) *o++ = d[i];
    }
    return (int)(o - buf);
}

// n numbers right-aligned to the widest one, separated by one space (Eigen's operator<< of a small vector), written at `o`
// (at least 32 n + n bytes); returns the end
inline char *fmt_vec_at(char *o, const double *v, int n)
{
    char buf[8][32];
    int len[8];
    if (n > 8) n = 8;  // the front end prints 2- and 3-vectors
    int w = 0;
    for (int i = 0; i < n; i++) {
        len[i] = fmt_g6(v[i], buf[i], sizeof(buf[i]));
        w = len[i] > w ? len[i] : w;
    }
    for (int i = 0; i < n; i++) {
        if (i) *o++ = ' ';
        for (int k = w - len[i]; k > 0; k--) *o++ = ' ';
        std::memcpy(o, buf[i], (size_t)len[i]);
        o += len[i];
    }
    return o;
}

inline void fmt_vec_append(std::string &out, const double *v, int n)
{
    char line[8 * 33];
    out.append(line, (size_t)(fmt_vec_at(line, v, n) - line));
}

}  // namespace vgtext
