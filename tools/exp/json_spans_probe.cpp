// json_spans_probe.cpp -- where the corner file's parse time goes on a given host: read, element_spans (serial), per-frame validation
// (parallel).  g++ -O2 -std=c++17 -pthread tools/exp/json_spans_probe.cpp -o /tmp/spans && /tmp/spans   (writes /tmp/vg_probe_corners.json)
#include "../../visgeom_amd/csrc/vg_json.hpp"
#include <chrono>
#include <cstdio>
#include <random>
int main()
{
    const char *path = "/tmp/vg_probe_corners.json";
    {
        std::mt19937_64 g(1);
        std::uniform_real_distribution<double> U(0, 1280);
        FILE *f = std::fopen(path, "w");
        std::fputc('[', f);
        for (int i = 0; i < 10000; i++) {
            std::fprintf(f, "%s[{\"camera\": \"cam\", \"points\": [", i ? ", " : "");
            for (int k = 0; k < 96; k++) std::fprintf(f, "%s[%.15g, %.15g]", k ? ", " : "", U(g), U(g));
            std::fputs("]}]", f);
        }
        std::fputc(']', f);
        std::fclose(f);
    }
    vgjson::TextFile t;
    for (int rep = 0; rep < 4; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        t.read(path);
        auto t1 = std::chrono::steady_clock::now();
        auto sp = vgjson::element_spans(t.c_str(), t.size());
        auto t2 = std::chrono::steady_clock::now();
        vgpar::parallel_ranges(sp.size(), 64, [&](size_t b, size_t e, int) {
            for (size_t f = b; f < e; f++) {
                vgjson::Cursor c(t.c_str(), sp[f].first, sp[f].second);
                c.skip();
            }
        });
        auto t3 = std::chrono::steady_clock::now();
        auto d = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
        std::printf("threads %d bytes %zu read %.1f spans %.1f (%zu) skip-validate %.1f ms\n", vgpar::host_threads(), t.size(), d(t0, t1), d(t1, t2), sp.size(), d(t2, t3));
    }
    std::remove(path);
}
