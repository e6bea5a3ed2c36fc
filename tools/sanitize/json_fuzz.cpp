// JSON jobs with random byte damage through the libimageflow ABI subset, under the sanitizers: the reader, the node
// dispatch up to the point where a device is needed, the error buffers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "imageflow_abi_subset.h"
#include "imageflow_hip.h"
int main() {
    srand(5);
    const std::string jobs[2] = {
        R"({"io":[{"io_id":0,"direction":"in","io":"placeholder"},{"io_id":1,"direction":"out","io":"output_buffer"}],"framewise":{"steps":[{"decode":{"io_id":0,"commands":[{"jpeg_downscale_hints":{"width":800,"height":600,"scale_luma_spatially":true}}]}},{"resample_2d":{"w":200,"h":200,"hints":{"down_filter":"robidoux","scaling_colorspace":"linear","sharpen_percent":15,"background_color":{"srgb":{"hex":"FFFFFFFF"}}}}},{"encode":{"io_id":1,"preset":{"libjpeg_turbo":{"quality":90,"progressive":true}}}}]}})",
        R"({"framewise":{"graph":{"nodes":{"0":{"create_canvas":{"w":64,"h":64,"format":"bgra_32","color":"transparent"}},"1":{"fill_rect":{"x1":0,"y1":0,"x2":10,"y2":10,"color":{"srgb":{"hex":"EECCFFFF"}}}},"2":{"constrain":{"mode":"within","w":32}},"3":{"command_string":{"kind":"ir4","value":"width=20&mode=max"}}},"edges":[{"from":0,"to":1,"kind":"input"},{"from":1,"to":2,"kind":"input"},{"from":2,"to":3,"kind":"input"}]}}})"};
    long answered = 0, errors = 0;
    const unsigned char jpeg_stub[19] = {0xFF, 0xD8, 0xFF};
    for (int j = 0; j < 2; ++j) {
        imageflow_context* c = imageflow_context_create(3, 2);
        if (!c) return 2;
        imageflow_context_add_input_buffer(c, 0, jpeg_stub, sizeof jpeg_stub, imageflow_lifetime_lifetime_outlives_context);
        imageflow_context_add_output_buffer(c, 1);
        for (int it = 0; it < 1500; ++it) {
            std::string m = jobs[j];
            const int muts = 1 + rand() % 6;
            for (int k = 0; k < muts; ++k) m[rand() % m.size()] = static_cast<char>(rand());
            if (rand() % 5 == 0) m.resize(1 + rand() % m.size());
            const imageflow_json_response* r = imageflow_context_send_json(c, (rand() & 1) ? "v1/execute" : "v1/build", reinterpret_cast<const uint8_t*>(m.data()), m.size());
            if (r) {
                int64_t status = 0; const uint8_t* buf = nullptr; size_t n = 0;
                imageflow_json_response_read(c, r, &status, &buf, &n);
                ++answered;
                imageflow_json_response_destroy(c, const_cast<imageflow_json_response*>(r));
            }
            if (imageflow_context_has_error(c)) { char msg[200]; size_t n = 0; imageflow_context_error_write_to_buffer(c, msg, sizeof msg, &n); ++errors; imageflow_context_error_try_clear(c); }
        }
        imageflow_context_destroy(c);
    }
    printf("answered %ld, errors %ld\n", answered, errors);
    // filter-weight tables (graphics/weights.rs populate_weights) for random shapes, every filter id and a few outside
    long tables = 0, refused = 0;
    std::vector<uint32_t> left(8192), count(8192);
    std::vector<float> w(1 << 20);
    for (int it = 0; it < 3000; ++it) {
        const int filter = rand() % 36 - 2, lobe = rand() % 4 - 1;
        const uint32_t out_n = 1 + rand() % 4000, in_n = 1 + rand() % 8000;
        uint32_t n = 0;
        const int rc = ifhip_populate_weights(filter, lobe, static_cast<float>(rand() % 200) / 100.0f, (rand() % 4) ? 1.0 : 0.2 + (rand() % 300) / 100.0,
                                              out_n, in_n, left.data(), count.data(), w.data(), static_cast<uint32_t>((rand() % 3) ? w.size() : rand() % 1000), &n);
        if (rc == 0) ++tables; else ++refused;
    }
    printf("weight tables %ld, refused %ld\n", tables, refused);
}
