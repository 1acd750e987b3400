// vg_json.hpp -- a small JSON reader for the calibration front end (the reference uses
// boost::property_tree, include/json.h:27-34; Boost is not available here).  Objects keep their key order;
// numbers are doubles; booleans may also be given as the strings "true"/"false" or as 0/1 the way
// property_tree's get<bool> accepts them.
#pragma once

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

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

inline Value parse_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    return Parser(text).parse();
}

}  // namespace vgjson
