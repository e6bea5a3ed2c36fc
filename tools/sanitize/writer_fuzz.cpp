#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include <cstring>
extern "C" int ifhip_jpeg_write(const int16_t*, const int16_t*, const int16_t*, const uint32_t*, const uint32_t*, int, const uint8_t*, const uint8_t*, uint32_t, uint32_t, int, int, uint8_t*, size_t, size_t*);
int main() {
    srand(7);
    int okc = 0, refused = 0;
    for (int iter = 0; iter < 400; ++iter) {
        const int ncomp = (rand() % 4) ? 3 : 1;
        const uint8_t hsv[3][2][3] = {{{1,1,1},{1,1,1}}, {{2,1,1},{1,1,1}}, {{2,1,1},{2,1,1}}};
        const int smp = rand() % 3;
        uint8_t hs[3], vs[3];
        for (int c = 0; c < 3; ++c) { hs[c] = ncomp == 3 ? hsv[smp][0][c] : 1; vs[c] = ncomp == 3 ? hsv[smp][1][c] : 1; }
        const uint32_t w = 1 + rand() % 70, h = 1 + rand() % 50;
        const uint32_t hmax = hs[0], vmax = vs[0];
        const uint32_t mw = (w + 8 * hmax - 1) / (8 * hmax), mh = (h + 8 * vmax - 1) / (8 * vmax);
        uint32_t bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
        std::vector<int16_t> pl[3];
        const int dens = rand() % 100, amp = 1 << (rand() % 11);
        for (int c = 0; c < ncomp; ++c) {
            bw[c] = mw * hs[c]; bh[c] = mh * vs[c];
            pl[c].resize((size_t)bw[c] * bh[c] * 64);
            for (auto& v : pl[c]) v = (rand() % 100 < dens) ? (int16_t)(rand() % (2 * amp + 1) - amp) : 0;
            if (rand() % 10 == 0) for (size_t i = 0; i < pl[c].size(); i += 64) pl[c][i] = (int16_t)(rand() % 2047 - 1023);
        }
        for (int flags = 0; flags < 4; ++flags) {
            std::vector<uint8_t> out(rand() % 2 ? 16 : (1 << 20));
            size_t len = 0;
            int rc = ifhip_jpeg_write(pl[0].data(), ncomp == 3 ? pl[1].data() : nullptr, ncomp == 3 ? pl[2].data() : nullptr, bw, bh, ncomp, hs, vs, w, h, 1 + rand() % 100, flags, out.data(), out.size(), &len);
            if (rc == 0) { ++okc; if (out[0] != 0xFF || out[1] != 0xD8 || out[len - 1] != 0xD9) { printf("bad file\n"); return 1; } }
            else ++refused;
        }
    }
    printf("ok %d refused %d\n", okc, refused);
}
