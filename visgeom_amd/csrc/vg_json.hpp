// vg_json.hpp -- a small JSON reader for the calibration front end (the reference uses
// boost::property_tree, include/json.h:27-34; Boost is not available here).  Objects keep their key order;
// numbers are doubles; booleans may also be given as the strings "true"/"false" or as 0/1 the way
// property_tree's get<bool> accepts them.
#pragma once

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cctype>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vg_host_parallel.hpp"

namespace vgjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool has(const std::string &key) const
    {
        for (auto &kv : obj)
            if (kv.first == key) return true;
        return false;
    }
    // "a.b" paths like property_tree's get_child("object.cols")
    const Value &at(const std::string &path) const
    {
        const size_t dot = path.find('.');
        const std::string head = path.substr(0, dot);
        for (auto &kv : obj)
            if (kv.first == head) return dot == std::string::npos ? kv.second : kv.second.at(path.substr(dot + 1));
        throw std::runtime_error("No such node (" + path + ")");
    }
    double as_number() const
    {
        if (kind == Number) return num;
        if (kind == String) {
            char *end = nullptr;
            const double v = std::strtod(str.c_str(), &end);
            if (end != str.c_str() && *end == 0) return v;
        }
        if (kind == Bool) return b ? 1. : 0.;
        throw std::runtime_error("conversion of data to number failed");
    }
    bool as_bool() const
    {
        if (kind == Bool) return b;
        if (kind == Number) return num != 0.;
        if (kind == String) {
            if (str == "true" || str == "1") return true;
            if (str == "false" || str == "0") return false;
        }
        throw std::runtime_error("conversion of data to bool failed");
    }
    const std::string &as_string() const
    {
        if (kind != String) throw std::runtime_error("conversion of data to string failed");
        return str;
    }
    std::vector<double> as_vector() const  // readVector<double>, include/json.h:69-78
    {
        std::vector<double> v;
        for (auto &x : arr) v.push_back(x.as_number());
        return v;
    }
};

class Parser {
public:
    explicit Parser(const std::string &text) : s(text) {}
    Value parse()
    {
        Value v = value();
        ws();
        if (i != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s;
    size_t i = 0;
    [[noreturn]] void fail(const std::string &m) const
    {
        throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": " + m);
    }
    void ws()
    {
        while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    }
    // recursion guard: a calibration file nests 6 levels deep; a hostile "[[[[..." must not exhaust the stack
    struct Depth {
        int &d;
        explicit Depth(int &depth) : d(depth) { d++; }
        ~Depth() { d--; }
    };
    int depth = 0;
    Value value()
    {
        Depth guard(depth);
        if (depth > 256) fail("nesting too deep");
        ws();
        if (i >= s.size()) fail("unexpected end");
        const char c = s[i];
        Value v;
        if (c == '{') {
            v.kind = Value::Object;
            i++;
            ws();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                i++;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Value::Array;
            i++;
            ws();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = Value::String;
            v.str = string();
            return v;
        }
        if (s.compare(i, 4, "true") == 0) { i += 4; v.kind = Value::Bool; v.b = true; return v; }
        if (s.compare(i, 5, "false") == 0) { i += 5; v.kind = Value::Bool; v.b = false; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
        char *end = nullptr;
        v.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) fail("unexpected character");
        i = (size_t)(end - s.c_str());
        v.kind = Value::Number;
        return v;
    }
    std::string string()
    {
        std::string out;
        i++;  // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                i++;
                if (i >= s.size()) fail("bad escape");
                switch (s[i]) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (i + 4 >= s.size()) fail("bad \\u escape");
                    const unsigned cp = (unsigned)std::strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16);
                    i += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += s[i];
                }
            } else {
                out += s[i];
            }
            i++;
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return out;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Streaming access for the bulk files (a 10 000-image corner file is 40 MB of text = 2 M numbers): the same grammar as
// Parser, read straight out of the text without building Values (the tree of such a file is 3 M nodes of 100 bytes;
// building it was 1.5 of the 1.7 s the front end spent on the file, profiles/NOTES.md round 5).  A Cursor validates
// what it skips; element_spans() cuts a top-level array into its elements so that they can be read by several threads.
class Cursor {
public:
    Cursor(const char *text, size_t begin, size_t end) : s(text), i(begin), n(end) {}
    size_t pos() const { return i; }
    const char *text() const { return s; }   // with pos(): a span to come back to (Cursor(text(), begin, end))
    bool at_end()
    {
        ws();
        return i >= n;
    }
    [[noreturn]] void fail(const std::string &m) const
    {
        throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": " + m);
    }
    void ws()
    {
        while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r' || s[i] == '\f' || s[i] == '\v')) i++;
    }
    char peek()
    {
        ws();
        if (i >= n) fail("unexpected end");
        return s[i];
    }
    // '[' ... : returns false for an empty array (the closing bracket is consumed)
    bool open(char bracket, char closing)
    {
        if (peek() != bracket) fail(std::string("expected '") + bracket + "'");
        i++;
        ws();
        if (i < n && s[i] == closing) {
            i++;
            return false;
        }
        return true;
    }
    // after an element: true = another one follows (the comma is consumed), false = the container is closed
    bool next(char closing)
    {
        ws();
        if (i < n && s[i] == ',') {
            i++;
            return true;
        }
        if (i < n && s[i] == closing) {
            i++;
            return false;
        }
        fail(std::string("expected ',' or '") + closing + "'");
    }
    void colon()
    {
        ws();
        if (i >= n || s[i] != ':') fail("expected ':'");
        i++;
    }
    std::string string()
    {
        if (peek() != '"') fail("expected a string");
        // the escapes are rare (camera names): take the slow path through Parser's routine only when one shows up
        size_t j = i + 1;
        while (j < n && s[j] != '"' && s[j] != '\\') j++;
        if (j < n && s[j] == '"') {
            std::string out(s + i + 1, j - i - 1);
            i = j + 1;
            return out;
        }
        std::string out;
        i++;
        while (i < n && s[i] != '"') {
            if (s[i] == '\\') {
                i++;
                if (i >= n) fail("bad escape");
                switch (s[i]) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (i + 4 >= n) fail("bad \\u escape");
                    const unsigned cp = (unsigned)std::strtoul(std::string(s + i + 1, 4).c_str(), nullptr, 16);
                    i += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += s[i];
                }
            } else {
                out += s[i];
            }
            i++;
        }
        if (i >= n) fail("unterminated string");
        i++;
        return out;
    }
    // a value used as a number: Value::as_number's conversions (a number, a numeric string, a boolean)
    double number()
    {
        const char c = peek();
        if (c == '"') {
            const std::string t = string();
            char *end = nullptr;
            const double v = std::strtod(t.c_str(), &end);
            if (end != t.c_str() && *end == 0) return v;
            throw std::runtime_error("conversion of data to number failed");
        }
        if (c == 't' && n - i >= 4 && std::memcmp(s + i, "true", 4) == 0) { i += 4; return 1.; }
        if (c == 'f' && n - i >= 5 && std::memcmp(s + i, "false", 5) == 0) { i += 5; return 0.; }
        if (c == '[' || c == '{' || c == 'n') throw std::runtime_error("conversion of data to number failed");
        return raw_number();
    }
    // validates and skips one value of any kind
    void skip(int depth = 0)
    {
        if (depth > 256) fail("nesting too deep");
        const char c = peek();
        if (c == '{') {
            if (!open('{', '}')) return;
            do {
                if (peek() != '"') fail("expected a key");
                (void)string();
                colon();
                skip(depth + 1);
            } while (next('}'));
        } else if (c == '[') {
            if (!open('[', ']')) return;
            do skip(depth + 1);
            while (next(']'));
        } else if (c == '"') {
            (void)string();
        } else if (c == 't' && n - i >= 4 && std::memcmp(s + i, "true", 4) == 0) {
            i += 4;
        } else if (c == 'f' && n - i >= 5 && std::memcmp(s + i, "false", 5) == 0) {
            i += 5;
        } else if (c == 'n' && n - i >= 4 && std::memcmp(s + i, "null", 4) == 0) {
            i += 4;
        } else {
            (void)raw_number();
        }
    }

private:
    const char *s;  // NUL-terminated behind `n` or beyond (std::string storage): strtod may look one character past a number
    size_t i, n;
    // decimal text -> double, correctly rounded (the value strtod returns), without calling into libc for the common shapes
    // (glibc's strtod does not scale over threads in this image -- 2 M conversions take 0.2 s on one thread and on eight,
    // profiles/NOTES.md round 5 -- and the corner files are 2 M numbers of 16-17 digits):
    //   * up to 15 significant digits and a decimal exponent within +-22: the digits and the power of ten are exact
    //     doubles, one IEEE multiplication or division rounds once (Clinger's fast path);
    //   * up to 19 digits (a 64-bit integer m) and a decimal exponent e within +-19: m * 10^e is an exact 128-bit integer
    //     (e >= 0), or the 128-bit quotient (m << s) / 10^-e with the remainder as a sticky bit carries >= 63 significant
    //     bits (e < 0); the integer -> double conversion rounds to nearest even once.  Values there lie in [1e-19, 2e38]:
    //     no subnormals, no overflow;
    //   * anything else (longer digit strings, large exponents, "inf", hex floats ...): strtod, as Parser does.
    double raw_number()
    {
        static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                          1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        static const unsigned long long kIntPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                                                         100000000ull, 1000000000ull, 10000000000ull, 100000000000ull,
                                                         1000000000000ull, 10000000000000ull, 100000000000000ull,
                                                         1000000000000000ull, 10000000000000000ull, 100000000000000000ull,
                                                         1000000000000000000ull, 10000000000000000000ull};
        const size_t start = i;
        size_t j = i;
        bool neg = false;
        if (j < n && (s[j] == '-' || s[j] == '+')) neg = s[j++] == '-';
        unsigned long long m = 0;
        int digits = 0, exp10 = 0;
        bool any = false, simple = true;
        auto digit = [&](char ch) {
            any = true;
            if (digits || ch != '0') {
                if (digits < 19) m = m * 10 + (unsigned)(ch - '0');
                else simple = false;  // a twentieth significant digit: strtod
                digits++;
            }
        };
        while (j < n && s[j] >= '0' && s[j] <= '9') digit(s[j++]);
        if (j < n && s[j] == '.') {
            j++;
            while (j < n && s[j] >= '0' && s[j] <= '9') {
                digit(s[j++]);
                exp10--;
            }
        }
        if (!any) simple = false;  // "inf", "nan", ".": whatever strtod makes of it, as Parser does
        if (simple && j < n && (s[j] == 'e' || s[j] == 'E')) {
            size_t k = j + 1;
            bool eneg = false;
            if (k < n && (s[k] == '-' || s[k] == '+')) eneg = s[k++] == '-';
            int e = 0;
            bool edig = false;
            while (k < n && s[k] >= '0' && s[k] <= '9') {
                edig = true;
                if (e < 10000) e = e * 10 + (s[k] - '0');
                k++;
            }
            if (edig) {
                exp10 += eneg ? -e : e;
                j = k;
            }
        }
        if (simple && j < n && (s[j] == 'x' || s[j] == 'X' || s[j] == 'p' || s[j] == 'P')) simple = false;  // hex float
        if (simple && m == 0) {
            i = j;
            return neg ? -0. : 0.;
        }
        if (simple && digits <= 15 && exp10 >= -22 && exp10 <= 22) {
            double v = (double)m;
            v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
            i = j;
            return neg ? -v : v;
        }
        if (simple && exp10 >= -19 && exp10 <= 19) {
            typedef unsigned __int128 u128;
            double v;
            if (exp10 >= 0) {
                v = (double)((u128)m * kIntPow10[exp10]);  // < 2^64 * 2^63.2: exact
            } else {
                const int shift = 63 + __builtin_clzll(m);  // m << shift has its leading bit at position 126
                const u128 num = (u128)m << shift;
                const u128 den = kIntPow10[-exp10];
                const u128 q = num / den;  // >= 2^126 / 2^63.2: at least 63 significant bits
                const u128 q2 = (q << 1) | (u128)(num % den != 0);  // the remainder as a sticky bit below them
                v = std::ldexp((double)q2, -(shift + 1));
            }
            i = j;
            return neg ? -v : v;
        }
        char *end = nullptr;
        const double v = std::strtod(s + start, &end);
        if (end == s + start) fail("unexpected character");
        i = (size_t)(end - s);
        if (i > n) fail("unexpected end");
        return v;
    }
};

// the [begin, end) spans of the elements of the top-level array of `text` (one structural pass: brackets outside strings)
inline std::vector<std::pair<size_t, size_t>> element_spans_serial(const char *s, const size_t n)
{
    std::vector<std::pair<size_t, size_t>> spans;
    Cursor c(s, 0, n);
    if (c.peek() != '[') c.fail("expected '['");
    size_t i = c.pos() + 1;
    int depth = 1;
    size_t start = std::string::npos;
    bool closed = false;
    // bytes that cannot change the state once an element has begun (digits, signs, letters, blanks ...): skipped in a tight loop
    static const std::vector<unsigned char> structural = [] {
        std::vector<unsigned char> t(256, 0);
        for (unsigned char ch : {'"', '[', ']', '{', '}', ','}) t[ch] = 1;
        return t;
    }();
    const unsigned char *special = structural.data();
    for (; i < n && !closed; i++) {
        if (start != std::string::npos)
            while (i < n && !special[(unsigned char)s[i]]) i++;
        if (i >= n) break;
        const char ch = s[i];
        switch (ch) {
        case '"': {
            if (depth == 1 && start == std::string::npos) start = i;
            size_t j = i + 1;
            for (;;) {
                const char *q = (const char *)std::memchr(s + j, '"', n - j);
                if (!q) throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": unterminated string");
                j = (size_t)(q - s);
                size_t b = j;
                while (b > i + 1 && s[b - 1] == '\\') b--;
                if (((j - b) & 1) == 0) break;  // an even number of backslashes in front: the quote closes the string
                j++;
            }
            i = j;
            break;
        }
        case '[':
        case '{':
            if (depth == 1 && start == std::string::npos) start = i;
            if (++depth > 257) throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": nesting too deep");
            break;
        case ']':
        case '}':
            if (--depth == 0) {
                if (ch != ']') throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": expected ',' or ']'");
                if (start != std::string::npos) spans.emplace_back(start, i);
                else if (!spans.empty()) throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": unexpected character");
                closed = true;
            }
            break;
        case ',':
            if (depth == 1) {
                if (start == std::string::npos) throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": unexpected character");
                spans.emplace_back(start, i);
                start = std::string::npos;
            }
            break;
        case ' ': case '\n': case '\t': case '\r': case '\f': case '\v': break;
        default:
            if (depth == 1 && start == std::string::npos) start = i;
        }
    }
    if (!closed) throw std::runtime_error("JSON parse error at offset " + std::to_string(n) + ": unexpected end");
    Cursor rest(s, i, n);
    if (!rest.at_end()) rest.fail("trailing characters");
    return spans;
}

// The same cut by several threads, for files of many megabytes (the serial pass over a 39 MB corner file is 13 ms on the GPU boxes'
// hosts, two thirds of its whole parse): three sweeps over ranges of the text side by side,
//   1. quotes that open or close a string (an even number of backslashes in front) per range -> which ranges START inside a string,
//   2. brackets outside strings per range -> the nesting depth at which each range starts,
//   3. the separators of the top level: the '[' that opens the array, every ',' at depth 1, the ']' that closes it.
// Anything but a well-formed cut (no opening bracket, an empty element, text behind the end, unbalanced brackets or strings, a nesting
// depth beyond the parser's limit) returns false: the caller then takes element_spans_serial, whose results and messages define
// the behaviour.
inline bool element_spans_parallel(const char *s, const size_t n, std::vector<std::pair<size_t, size_t>> &spans)
{
    const size_t P = (size_t)vgpar::host_threads();
    if (P < 2) return false;
    auto chunk_begin = [&](size_t k) { return n * k / P; };
    auto real_quote = [&](size_t j) {   // s[j] == '"': does it open / close a string?
        size_t b = j;
        while (b > 0 && s[b - 1] == '\\') b--;
        return ((j - b) & 1) == 0;
    };
    // 1. quote parity per range
    std::vector<unsigned char> odd(P, 0), in_string(P + 1, 0);
    vgpar::parallel_ranges(P, 1, [&](size_t kb, size_t ke, int) {
        for (size_t k = kb; k < ke; k++) {
            const size_t e = chunk_begin(k + 1);
            size_t j = chunk_begin(k);
            unsigned char par = 0;
            while (j < e) {
                const char *q = (const char *)std::memchr(s + j, '"', e - j);
                if (!q) break;
                j = (size_t)(q - s);
                if (real_quote(j)) par ^= 1;
                j++;
            }
            odd[k] = par;
        }
    });
    for (size_t k = 0; k < P; k++) in_string[k + 1] = in_string[k] ^ odd[k];
    if (in_string[P]) return false;   // unterminated string
    // a sweep over one range outside strings: on_bracket(pos, ch) for [ ] { }, on_comma(pos) for ','
    // (a backslash outside a string is no JSON, and the quote test above would read it as an escape: such a text is declined)
    static const std::vector<unsigned char> structural = [] {
        std::vector<unsigned char> t(256, 0);
        for (unsigned char ch : {'"', '[', ']', '{', '}', ',', '\\'}) t[ch] = 1;
        return t;
    }();
    const unsigned char *cls = structural.data();
    auto sweep = [&](size_t k, auto &&on_bracket, auto &&on_comma) {
        const size_t e = chunk_begin(k + 1);
        size_t j = chunk_begin(k);
        bool str = in_string[k] != 0;
        while (j < e) {
            if (str) {   // to the quote that closes the string
                const char *q = (const char *)std::memchr(s + j, '"', e - j);
                if (!q) return;
                j = (size_t)(q - s);
                if (real_quote(j)) str = false;
                j++;
                continue;
            }
            while (j < e && !cls[(unsigned char)s[j]]) j++;   // digits, signs, letters, blanks
            if (j >= e) return;
            const char ch = s[j];
            if (ch == '"') str = true;
            else if (ch == ',') on_comma(j);
            else on_bracket(j, ch);
            j++;
        }
    };
    // 2. depth at the start of each range
    std::vector<long long> delta(P, 0), depth0(P + 1, 0);
    std::vector<unsigned char> stray(P, 0);
    vgpar::parallel_ranges(P, 1, [&](size_t kb, size_t ke, int) {
        for (size_t k = kb; k < ke; k++) {
            long long d = 0;
            sweep(
                k,
                [&](size_t, char ch) {
                    if (ch == '\\') stray[k] = 1;
                    else d += (ch == '[' || ch == '{') ? 1 : -1;
                },
                [](size_t) {});
            delta[k] = d;
        }
    });
    for (size_t k = 0; k < P; k++) {
        if (stray[k]) return false;
        depth0[k + 1] = depth0[k] + delta[k];
    }
    if (depth0[P] != 0) return false;
    // 3. separators of the top level, in text order: (position, kind) with kind 0 = array opens, 1 = comma, 2 = array closes
    std::vector<std::vector<std::pair<size_t, int>>> seps(P);
    std::vector<unsigned char> odd_shape(P, 0);
    vgpar::parallel_ranges(P, 1, [&](size_t kb, size_t ke, int) {
        for (size_t k = kb; k < ke; k++) {
            long long d = depth0[k];
            auto &out = seps[k];
            sweep(
                k,
                [&](size_t j, char ch) {
                    if (ch == '\\') {
                        odd_shape[k] = 1;   // (the second sweep has already declined such a text)
                    } else if (ch == '[' || ch == '{') {
                        if (d == 0) {
                            if (ch != '[') odd_shape[k] = 1;
                            out.emplace_back(j, 0);
                        }
                        if (++d > 257) odd_shape[k] = 1;
                    } else {
                        if (--d == 0) {
                            if (ch != ']') odd_shape[k] = 1;
                            out.emplace_back(j, 2);
                        }
                        if (d < 0) odd_shape[k] = 1;
                    }
                },
                [&](size_t j) {
                    if (d == 1) out.emplace_back(j, 1);
                    else if (d <= 0) odd_shape[k] = 1;
                });
        }
    });
    size_t total = 0;
    for (size_t k = 0; k < P; k++) {
        if (odd_shape[k]) return false;
        total += seps[k].size();
    }
    if (total < 2) return false;
    auto is_ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; };
    spans.clear();
    spans.reserve(total);
    size_t prev = 0, seen = 0;
    for (size_t k = 0; k < P; k++)
        for (const auto &sp : seps[k]) {
            const size_t j = sp.first;
            const int kind = sp.second;
            if (seen == 0) {   // the array must open first, behind blanks only
                if (kind != 0) return false;
                for (size_t i = 0; i < j; i++)
                    if (!is_ws(s[i])) return false;
            } else {
                if (kind == 0) return false;   // a second top-level value
                size_t a = prev + 1;
                while (a < j && is_ws(s[a])) a++;
                if (a == j) {   // nothing between two separators: only "[]" may do that
                    if (!(kind == 2 && seen == 1)) return false;
                } else {
                    spans.emplace_back(a, j);
                }
                if (kind == 2) {   // closed: blanks only behind it, and it must be the last separator
                    if (seen + 1 != total) return false;
                    for (size_t i = j + 1; i < n; i++)
                        if (!is_ws(s[i])) return false;
                    return true;
                }
            }
            prev = j;
            seen++;
        }
    return false;   // never closed
}

inline std::vector<std::pair<size_t, size_t>> element_spans(const char *s, const size_t n)
{
    std::vector<std::pair<size_t, size_t>> spans;
    if (n >= ((size_t)4 << 20) && element_spans_parallel(s, n, spans)) return spans;
    return element_spans_serial(s, n);
}
inline std::vector<std::pair<size_t, size_t>> element_spans(const std::string &text) { return element_spans(text.c_str(), text.size()); }

inline std::string read_text_file(const std::string &path)
{
    std::string text;
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    if (n > 0) {
        text.resize((size_t)n);
        f.read(&text[0], n);
        text.resize((size_t)f.gcount());
    } else {  // not seekable: stream it
        std::stringstream ss;
        ss << f.rdbuf();
        text = ss.str();
    }
    return text;
}

// A whole file in memory, zero-terminated.  Regular files are read as ranges side by side by the host's threads into a block
// that nobody clears first (a 39 MB corner file: 11 ms through one stream into a zero-filled string, of which the zero fill and
// the page faults of one thread were half); anything else (a pipe, /dev/stdin) through the stream.
class TextFile {
    std::unique_ptr<char[]> buf_;
    std::string fallback_;
    size_t size_ = 0;

public:
    const char *c_str() const { return buf_ ? buf_.get() : fallback_.c_str(); }
    size_t size() const { return size_; }
    void read(const std::string &path)
    {
        buf_.reset();
        const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) {
            ::close(fd);
            fallback_ = read_text_file(path);
            size_ = fallback_.size();
            return;
        }
        const size_t n = (size_t)st.st_size;
        buf_.reset(new char[n + 1]);
        char *dst = buf_.get();
        std::atomic<bool> failed(false);
        vgpar::parallel_ranges(n, (size_t)4 << 20, [&](size_t b, size_t e, int) {
            size_t done = b;
            while (done < e) {
                const ssize_t r = ::pread(fd, dst + done, e - done, (off_t)done);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) break;   // error, or the file shrank under us
                done += (size_t)r;
            }
            if (done != e) failed.store(true);
        });
        ::close(fd);
        if (failed.load()) throw std::runtime_error("cannot read " + path);
        dst[n] = 0;
        size_ = n;
    }
};

// read_seconds / parse_seconds / bytes (each may be NULL) are ADDED to: the front end's phase clock
inline Value parse_file(const std::string &path, double *read_seconds = nullptr, double *parse_seconds = nullptr, int64_t *bytes = nullptr)
{
    const auto t0 = std::chrono::steady_clock::now();
    const std::string text = read_text_file(path);
    const auto t1 = std::chrono::steady_clock::now();
    Value v = Parser(text).parse();
    const auto t2 = std::chrono::steady_clock::now();
    if (read_seconds) *read_seconds += std::chrono::duration<double>(t1 - t0).count();
    if (parse_seconds) *parse_seconds += std::chrono::duration<double>(t2 - t1).count();
    if (bytes) *bytes += (int64_t)text.size();
    return v;
}

}  // namespace vgjson
