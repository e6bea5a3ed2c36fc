#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "imageflow_hip.h"
int main(int argc, char** argv) {
    srand(3);
    long tried = 0, parsed = 0, reported = 0;
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
        std::vector<uint8_t> d(n); if (fread(d.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
        for (int it = 0; it < 300; ++it) {
            std::vector<uint8_t> m = d;
            const int muts = rand() % 6;
            for (int k = 0; k < muts; ++k) m[rand() % m.size()] = (uint8_t)rand();
            if (rand() % 6 == 0) m.resize(4 + rand() % (m.size() - 4));
            uint32_t w, h, bw[3], bh[3], ri; int nc; uint8_t hs[3], vs[3]; uint16_t qt[192];
            ++tried;
            if (ifhip_jpeg_parse_headers(m.data(), m.size(), &w, &h, &nc, hs, vs, bw, bh, qt, &ri) == 0) ++parsed;
            ifhip_jpeg_scan_report rep;
            if (ifhip_jpeg_debug_scan_report(m.data(), m.size(), &rep) == 0) ++reported;
        }
    }
    printf("tried %ld parsed %ld reported %ld\n", tried, parsed, reported);
}
