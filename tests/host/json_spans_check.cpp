// json_spans_check.cpp -- vgjson::element_spans_parallel against element_spans_serial (visgeom_amd/csrc/vg_json.hpp) -- test infrastructure.
// Random top-level arrays whose strings carry brackets, commas, quotes and backslash runs, cut into as many ranges as there are host
// threads (the documents are short, so range boundaries fall inside strings, escapes and numbers all the time), and damaged copies of
// them: the parallel cut must either equal the serial one or decline (return false) -- it must decline whenever the serial one throws.
// Prints "checked N bad B"; exit code 1 on any difference.  Built and run by tests/test_frontend_cpu.py.
#include <cstdio>
#include <random>
#include <string>

#include "../../visgeom_amd/csrc/vg_json.hpp"

static std::mt19937_64 g(20260929);
static int rnd(int n) { return (int)(g() % (unsigned long long)n); }

static std::string rand_string()
{
    static const char *pieces[] = {"a", "]", "[", "{", "}", ",", "\\\"", "\\\\", "\\\\\\\"", " ", "x\\\\", "\\n", "0.5", ":"};
    std::string s = "\"";
    for (int k = rnd(6); k > 0; k--) s += pieces[rnd(14)];
    return s + "\"";
}

static std::string rand_ws()
{
    static const char *w[] = {"", "", " ", "\n", "\t ", "  "};
    return w[rnd(6)];
}

static std::string rand_value(int depth)
{
    const int t = rnd(depth > 3 ? 3 : 6);
    if (t == 0) return std::to_string(rnd(2000) / 8.0);
    if (t == 1) return rand_string();
    if (t == 2) return rnd(2) ? "true" : "null";
    if (t <= 4) {
        std::string s = "[" + rand_ws();
        for (int k = rnd(4), i = 0; i < k; i++) s += (i ? "," + rand_ws() : "") + rand_value(depth + 1) + rand_ws();
        return s + "]";
    }
    std::string s = "{" + rand_ws();
    for (int k = rnd(3), i = 0; i < k; i++) s += (i ? "," : "") + rand_ws() + rand_string() + rand_ws() + ":" + rand_ws() + rand_value(depth + 1);
    return s + rand_ws() + "}";
}

static long long checked = 0, bad = 0, declined_good = 0;

static void check(const std::string &doc, bool pristine = false)
{
    std::vector<std::pair<size_t, size_t>> par, ser;
    bool threw = false;
    try {
        ser = vgjson::element_spans_serial(doc.c_str(), doc.size());
    } catch (const std::exception &) {
        threw = true;
    }
    const bool ok = vgjson::element_spans_parallel(doc.c_str(), doc.size(), par);
    checked++;
    if (ok && (threw || par != ser)) {
        if (bad < 10) std::printf("MISMATCH (serial %s) on: %s\n", threw ? "throws" : "differs", doc.c_str());
        bad++;
    }
    if (!ok && !threw && pristine) {   // (a damaged document that still balances may be declined: the serial cut then decides)
        if (declined_good < 8) std::printf("DECLINED: %s\n", doc.c_str());
        declined_good++;
    }
}

int main()
{
    for (int it = 0; it < 4000; it++) {
        std::string doc = rand_ws() + "[" + rand_ws();
        for (int k = rnd(8), i = 0; i < k; i++) doc += (i ? "," + rand_ws() : "") + rand_value(0) + rand_ws();
        doc += "]" + rand_ws();
        check(doc, true);
        // damaged copies: a byte removed, a byte replaced by a structural one, a truncation, a tail
        if (!doc.empty()) {
            std::string d1 = doc;
            d1.erase((size_t)rnd((int)d1.size()), 1);
            check(d1);
            std::string d2 = doc;
            d2[(size_t)rnd((int)d2.size())] = "[]{},\"\\ "[rnd(8)];
            check(d2);
            check(doc.substr(0, (size_t)rnd((int)doc.size())));
            check(doc + ",[1]");
            check(doc + rand_value(0));
        }
    }
    check("");
    check("[]");
    check(" [ ] ");
    check("[,]");
    check("[1,]");
    check("[1 2]");
    check("{}");
    check("[1]]");
    check("[\"a]");
    // an undamaged document must take the fast path
    std::printf("threads %d declined well-formed %lld\n", vgpar::host_threads(), declined_good);
    std::printf("checked %lld bad %lld\n", checked, bad);
    return bad || declined_good ? 1 : 0;
}
