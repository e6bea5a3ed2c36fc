#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "imageflow_hip.h"
namespace ifhip { const char* last_error(); }   // (tools/sanitize/stubs.cpp stands in for api.cpp here)
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
            const int rrc = ifhip_jpeg_debug_scan_report(m.data(), m.size(), &rep);
            if (rrc == 0) ++reported;
            else if (strstr(ifhip::last_error(), "packed un-stuffer disagrees")) { printf("FAIL: %s\n", ifhip::last_error()); return 1; }
        }
    }
    printf("tried %ld parsed %ld reported %ld\n", tried, parsed, reported);
    // the metadata readers (EXIF orientation, ICC profile kind) on files that carry such segments, the segments mutated:
    // lengths, counts, offsets and tag tables are all attacker bytes
    long meta = 0, kinds[3] = {0, 0, 0};
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
        std::vector<uint8_t> d(n); if (fread(d.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
        for (int it = 0; it < 400; ++it) {
            // an ICC-shaped profile: header, a tag table whose offsets / sizes are partly random, some plausible tags
            std::vector<uint8_t> icc(128 + 4 + 12 * 6 + 200 + rand() % 300, 0);
            const uint32_t sz = (uint32_t)icc.size();
            icc[0] = sz >> 24; icc[1] = sz >> 16; icc[2] = sz >> 8; icc[3] = sz;
            memcpy(&icc[16], (rand() % 5) ? "RGB " : "CMYK", 4); memcpy(&icc[20], "XYZ ", 4);
            icc[131] = (uint8_t)(rand() % 9);
            const char* sigs[6] = {"rXYZ", "gXYZ", "bXYZ", "rTRC", "gTRC", "bTRC"};
            for (int t = 0; t < 6; ++t) {
                uint8_t* e = &icc[132 + 12 * t];
                memcpy(e, sigs[t], 4);
                uint32_t off = 204 + (rand() % 4 ? 20 * t : rand()), size = rand() % 3 ? 32 : (uint32_t)rand();
                e[4] = off >> 24; e[5] = off >> 16; e[6] = off >> 8; e[7] = off; e[8] = size >> 24; e[9] = size >> 16; e[10] = size >> 8; e[11] = size;
            }
            for (size_t k = 204; k + 4 <= icc.size(); k += 20) memcpy(&icc[k], (k / 20) % 2 ? "XYZ " : ((rand() & 1) ? "para" : "curv"), 4);
            for (int k = rand() % 8; k > 0; --k) icc[rand() % icc.size()] = (uint8_t)rand();
            // one to three APP2 chunks with sequence numbers that are sometimes wrong, and an Exif APP1 with random IFD bytes
            std::vector<uint8_t> m(d.begin(), d.begin() + 2);
            const int pieces = 1 + rand() % 3;
            const size_t step = (icc.size() + pieces - 1) / pieces;
            for (int k = 0; k < pieces; ++k) {
                const size_t a0 = k * step, a1 = a0 + step < icc.size() ? a0 + step : icc.size();
                const size_t len = 2 + 14 + (a1 - a0);
                m.push_back(0xFF); m.push_back(0xE2); m.push_back((uint8_t)(len >> 8)); m.push_back((uint8_t)len);
                const char* id = "ICC_PROFILE"; m.insert(m.end(), id, id + 12);
                m.push_back((uint8_t)((rand() % 7) ? k + 1 : rand())); m.push_back((uint8_t)((rand() % 7) ? pieces : rand()));
                m.insert(m.end(), icc.begin() + a0, icc.begin() + a1);
            }
            std::vector<uint8_t> ex(40 + rand() % 40);
            for (auto& b : ex) b = (uint8_t)rand();
            memcpy(ex.data(), "Exif\0\0II\x2a\0", 10);
            m.push_back(0xFF); m.push_back(0xE1); m.push_back(0); m.push_back((uint8_t)(ex.size() + 2));
            m.insert(m.end(), ex.begin(), ex.end());
            m.insert(m.end(), d.begin() + 2, d.end());
            if (rand() % 5 == 0) m.resize(20 + rand() % 600);
            int kind = -1, flag = -2;
            if (ifhip_jpeg_icc_profile_kind(m.data(), m.size(), &kind) == 0 && kind >= 0 && kind <= 2) ++kinds[kind];
            (void)ifhip_jpeg_exif_orientation(m.data(), m.size(), &flag);
            ++meta;
        }
    }
    printf("metadata readers: %ld files, ICC kinds none/srgb/other %ld/%ld/%ld\n", meta, kinds[0], kinds[1], kinds[2]);
}
