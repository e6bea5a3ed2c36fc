// abi_shim.cpp -- the OUTER drop-in boundary: a subset of libimageflow's C ABI v3.2 (imageflow_abi/src/lib.rs:389-1496,
// header bindings/headers/imageflow_default.h) over the gfx950 kernels, declared in include/imageflow_abi_subset.h.
//
// Deliberately thin: a context with an io table, a sticky error and a response list; a small
// JSON reader; and a straight-line interpreter for the `v1/build` / `v1/execute` job shapes that reach the pixel hot
// path -- decode(baseline JPEG) -> [orientation / crop / canvas primitives] -> resample_2d | constrain | command_string
// -> encode -- as `steps` or as a `graph` with `input` edges (one producer per node, any number of consumers).  It is
// NOT imageflow's router or graph engine: nodes outside that list answer ActionNotSupported, and everything a job
// computes is computed by the ifhip_* entry points of this library on frames that stay in HBM.
//
// Two labelled EXTENSIONS, because the reference's JSON API has no raw-pixel I/O:
//   * decode accepts, besides baseline JPEG, the container "IFBGRA1\0" + u32le w, h, stride, alpha_meaningful + rows;
//   * encode writes a real JPEG for the libjpeg_turbo preset -- baseline, optimised tables, progressive (device pixel stage + host Huffman coder, jpeg_write.cpp)
//     and that container for every other preset (preferred_extension "ifbgra", mime "application/x-imageflow-bgra"):
//     PNG deflate / GIF / WebP coders are out of scope, the caller's encoder takes the frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/imageflow_abi_subset.h"
#include "../../include/imageflow_hip.h"

namespace {

// ---- errors: ErrorCategory (imageflow_core/src/errors.rs:779-838) with its exit / HTTP maps (:849-902) -----------
enum Cat { kOk = 0, kOutOfMemory = 1, kArgumentInvalid = 2, kInvalidJson = 3, kImageMalformed = 4, kImageTypeNotSupported = 5,
           kNodeArgumentInvalid = 6, kGraphInvalid = 7, kActionNotSupported = 8, kIoError = 16, kInternalError = 18 };
int http_code(int c) {
    switch (c) {
    case kOk: return 200;
    case kArgumentInvalid: case kGraphInvalid: case kNodeArgumentInvalid: case kActionNotSupported: case kInvalidJson:
    case kImageMalformed: case kImageTypeNotSupported: return 400;
    case kOutOfMemory: return 503;
    default: return 500;
    }
}
int exit_code(int c) {
    switch (c) {
    case kOk: return 0;
    case kArgumentInvalid: case kGraphInvalid: case kActionNotSupported: case kNodeArgumentInvalid: return 64;
    case kInvalidJson: case kImageMalformed: case kImageTypeNotSupported: return 65;
    case kOutOfMemory: return 71;
    case kIoError: return 74;
    default: return 70;
    }
}
struct FlowErr {
    int cat;
    std::string msg;
};
[[noreturn]] void raise(int cat, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void raise(int cat, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw FlowErr{cat, buf};
}
void check(int rc) {                                  // ifhip_status -> ErrorCategory
    if (rc == IFHIP_OK) return;
    const char* m = ifhip_last_error_message();
    const std::string msg = m ? m : "";
    int cat = kInternalError;
    if (rc == IFHIP_INVALID_ARGUMENT) cat = msg.rfind("ImageMalformed", 0) == 0 ? kImageMalformed : kArgumentInvalid;
    else if (rc == IFHIP_METHOD_NOT_IMPLEMENTED) cat = kActionNotSupported;
    else if (rc == IFHIP_ALLOCATION_FAILED) cat = kOutOfMemory;
    throw FlowErr{cat, msg};
}
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) raise(e == hipErrorOutOfMemory ? kOutOfMemory : kInternalError, "GpuError: %s: %s", what, hipGetErrorString(e));
}

// ---- JSON ------------------------------------------------------------------------------------------------------
struct JVal {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    double n = 0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* k) const {
        if (t != Obj) return nullptr;
        for (const auto& kv : o) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool is_null() const { return t == Null; }
};
struct JParser {
    const char *p, *end;
    int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    [[noreturn]] void bad(const char* what) { raise(kInvalidJson, "InvalidJson: %s at byte %ld", what, static_cast<long>(end - p)); }
    void lit(const char* w) { const size_t n = std::strlen(w); if (static_cast<size_t>(end - p) < n || std::memcmp(p, w, n)) bad("bad literal"); p += n; }
    std::string str() {
        std::string out;
        ++p;
        while (true) {
            if (p >= end) bad("unterminated string");
            const unsigned char c = static_cast<unsigned char>(*p++);
            if (c == '"') return out;
            if (c < 0x20) bad("control character in string");
            if (c != '\\') { out.push_back(static_cast<char>(c)); continue; }
            if (p >= end) bad("unterminated escape");
            const char e = *p++;
            switch (e) {
            case '"': case '\\': case '/': out.push_back(e); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                if (end - p < 4) bad("short \\u escape");
                unsigned v = 0;
                for (int i = 0; i < 4; ++i) {
                    const char h = *p++;
                    v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : (bad("bad hex digit"), 0));
                }
                if (v < 0x80) out.push_back(static_cast<char>(v));
                else if (v < 0x800) { out.push_back(static_cast<char>(0xC0 | (v >> 6))); out.push_back(static_cast<char>(0x80 | (v & 63))); }
                else { out.push_back(static_cast<char>(0xE0 | (v >> 12))); out.push_back(static_cast<char>(0x80 | ((v >> 6) & 63))); out.push_back(static_cast<char>(0x80 | (v & 63))); }
                break;
            }
            default: bad("bad escape");
            }
        }
    }
    JVal value() {
        if (++depth > 64) bad("nesting too deep");
        ws();
        if (p >= end) bad("unexpected end");
        JVal v;
        const char c = *p;
        if (c == '{') {
            v.t = JVal::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; --depth; return v; }
            while (true) {
                ws();
                if (p >= end || *p != '"') bad("expected a key");
                std::string k = str();
                ws();
                if (p >= end || *p != ':') bad("expected ':'");
                ++p;
                v.o.emplace_back(std::move(k), value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                bad("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.t = JVal::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; --depth; return v; }
            while (true) {
                v.a.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                bad("expected ',' or ']'");
            }
        } else if (c == '"') { v.t = JVal::Str; v.s = str(); }
        else if (c == 't') { lit("true"); v.t = JVal::Bool; v.b = true; }
        else if (c == 'f') { lit("false"); v.t = JVal::Bool; }
        else if (c == 'n') { lit("null"); }
        else {
            char* e = nullptr;
            const std::string tmp(p, static_cast<size_t>(std::min<long>(end - p, 64)));
            v.n = std::strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) bad("unexpected character");
            p += e - tmp.c_str();
            v.t = JVal::Num;
        }
        --depth;
        return v;
    }
};
JVal parse_json(const uint8_t* buf, size_t n) {
    JParser P{reinterpret_cast<const char*>(buf), reinterpret_cast<const char*>(buf) + n};
    JVal v = P.value();
    P.ws();
    if (P.p != P.end) P.bad("trailing characters");
    return v;
}
int64_t want_int(const JVal& o, const char* key, const char* node) {
    const JVal* v = o.get(key);
    if (!v || v->t != JVal::Num || v->n != std::floor(v->n)) raise(kInvalidJson, "InvalidJson: %s.%s must be an integer", node, key);
    return static_cast<int64_t>(v->n);
}
uint32_t want_u32(const JVal& o, const char* key, const char* node) {
    const int64_t v = want_int(o, key, node);
    if (v < 0 || v > 0x7fffffff) raise(kInvalidJson, "InvalidJson: %s.%s out of range", node, key);
    return static_cast<uint32_t>(v);
}

// imageflow_types::Color (lib.rs:807-824) + imageflow_helpers/src/colors.rs:36-61 -> Color32 0xAARRGGBB
uint32_t parse_color(const JVal* v, const char* node) {
    if (!v || v->is_null()) return 0u;
    if (v->t == JVal::Str) {
        if (v->s == "transparent") return 0u;
        if (v->s == "black") return 0xFF000000u;
        raise(kInvalidJson, "InvalidJson: %s: unknown colour '%s'", node, v->s.c_str());
    }
    const JVal* srgb = v->get("srgb");
    const JVal* hex = srgb ? srgb->get("hex") : nullptr;
    if (!hex || hex->t != JVal::Str) raise(kInvalidJson, "InvalidJson: %s: colour must be \"transparent\", \"black\" or {\"srgb\":{\"hex\":..}}", node);
    std::string s = hex->s;
    if (!s.empty() && s[0] == '#') s.erase(0, 1);
    if (s.size() == 3 || s.size() == 4) { std::string d; for (char c : s) { d.push_back(c); d.push_back(c); } s = d; }
    if (s.size() == 6) s += "FF";
    if (s.size() != 8) raise(kNodeArgumentInvalid, "InvalidNodeParams: %s: bad colour '%s'", node, hex->s.c_str());
    uint32_t ch[4];
    for (int i = 0; i < 4; ++i) {
        char* e = nullptr;
        const std::string part = s.substr(static_cast<size_t>(2 * i), 2);
        ch[i] = static_cast<uint32_t>(std::strtoul(part.c_str(), &e, 16));
        if (e != part.c_str() + 2) raise(kNodeArgumentInvalid, "InvalidNodeParams: %s: bad colour '%s'", node, hex->s.c_str());
    }
    return (ch[3] << 24) | (ch[0] << 16) | (ch[1] << 8) | ch[2];
}
int parse_filter(const JVal* v, int dflt) {                      // imageflow_types/src/lib.rs:144-205
    if (!v || v->is_null()) return dflt;
    static const std::pair<const char*, int> names[] = {
        {"robidoux_fast", 1}, {"robidoux", 2}, {"robidoux_sharp", 3}, {"ginseng", 4}, {"ginseng_sharp", 5}, {"lanczos", 6},
        {"lanczos_sharp", 7}, {"lanczos_2", 8}, {"lanczos_2_sharp", 9}, {"cubic", 11}, {"cubic_sharp", 12}, {"catmull_rom", 13},
        {"mitchell", 14}, {"cubic_b_spline", 15}, {"hermite", 16}, {"jinc", 17}, {"triangle", 22}, {"linear", 23}, {"box", 24},
        {"fastest", 27}, {"n_cubic", 29}, {"n_cubic_sharp", 30}};
    if (v->t == JVal::Str)
        for (const auto& kv : names) if (v->s == kv.first) return kv.second;
    raise(kInvalidJson, "InvalidJson: unknown filter");
}

// ---- frames in HBM ---------------------------------------------------------------------------------------------
struct Frame {                                   // graphics/bitmaps.rs Bitmap: BGRA8, 64-byte row stride
    uint8_t* d = nullptr;
    uint32_t w = 0, h = 0, stride = 0;
    bool alpha = false;
    int compose = IFHIP_REPLACE_SELF;
    uint32_t matte = 0;
    size_t bytes() const { return static_cast<size_t>(h) * stride; }
    ~Frame() { if (d) (void)hipFree(d); }
};
using FramePtr = std::shared_ptr<Frame>;
FramePtr new_frame(uint32_t w, uint32_t h, bool alpha, uint32_t fill_color32 = 0, bool zero = true) {
    if (w == 0 || h == 0) raise(kArgumentInvalid, "InvalidArgument: Bitmap dimensions cannot be zero");
    auto f = std::make_shared<Frame>();
    f->w = w; f->h = h; f->stride = ifhip_stride_for_width(w); f->alpha = alpha;
    hip_check(hipMalloc(reinterpret_cast<void**>(&f->d), f->bytes() + 64), "hipMalloc(frame)");
    if (zero) hip_check(hipMemsetAsync(f->d, 0, f->bytes() + 64, nullptr), "hipMemset(frame)");
    if (fill_color32 >> 24) {                    // create_canvas.rs:77-103 / bitmaps.rs:829-837: matte canvases start filled
        f->compose = IFHIP_BLEND_WITH_MATTE; f->matte = fill_color32;
        check(ifhip_fill_rect_batch_device(f->d, f->bytes(), 1, w, h, f->stride, IFHIP_REPLACE_SELF, 0, 0, w, h, fill_color32, nullptr));
    }
    return f;
}

constexpr char kRawMagic[8] = {'I', 'F', 'B', 'G', 'R', 'A', '1', '\0'};
constexpr size_t kRawHeader = 8 + 16;

// ---- context -----------------------------------------------------------------------------------------------------
struct Io {
    bool is_output = false;
    const uint8_t* in = nullptr;
    size_t in_len = 0;
    std::vector<uint8_t> owned;                  // copied inputs (lifetime_outlives_function_call) / output bytes
    bool written = false;
};
struct Response {
    int64_t status;
    std::string json;
};
}  // namespace

struct imageflow_json_response {                 // opaque to callers; read through imageflow_json_response_read
    Response r;
};
struct imageflow_context {
    std::mutex mu;
    std::map<int32_t, Io> io;
    int err_cat = kOk;
    std::string err_msg;
    std::vector<std::unique_ptr<imageflow_json_response>> responses;
    std::vector<std::unique_ptr<uint8_t[]>> allocations;
    void set_error(int cat, const std::string& m) { if (err_cat == kOk) { err_cat = cat; err_msg = m; } }   // first error sticks
};

namespace {

std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(static_cast<char>(c)); }
        else if (c == '\n') o += "\\n";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back(static_cast<char>(c));
    }
    return o;
}
const imageflow_json_response* respond(imageflow_context* c, int64_t status, const std::string& json) {
    c->responses.emplace_back(new imageflow_json_response{Response{status, json}});
    return c->responses.back().get();
}
const imageflow_json_response* respond_error(imageflow_context* c, int cat, const std::string& msg) {
    c->set_error(cat, msg);                                      // imageflow_abi/src/lib.rs:1001-1008
    const int code = http_code(cat);                             // JsonResponse::fail_with_message, json/mod.rs:170-181
    return respond(c, code, "{\n  \"code\": " + std::to_string(code) + ",\n  \"success\": false,\n  \"message\": \"" +
                                json_escape(msg) + "\",\n  \"data\": {\n    \"none\": null\n  }\n}");
}

// ---- the job interpreter ---------------------------------------------------------------------------------------
struct EncodeRecord { int32_t io_id; uint32_t w, h; const char* mime; const char* ext; };
struct DecodeRecord { int32_t io_id; uint32_t w, h; const char* mime; const char* ext; };
struct Job {
    imageflow_context* c;
    std::vector<EncodeRecord> encodes;
    std::vector<DecodeRecord> decodes;

    Io& input(int32_t id) {
        auto it = c->io.find(id);
        if (it == c->io.end() || it->second.is_output) raise(kArgumentInvalid, "InvalidArgument: io_id %d is not a registered input", id);
        return it->second;
    }
    Io& output(int32_t id) {
        auto it = c->io.find(id);
        if (it == c->io.end() || !it->second.is_output) raise(kArgumentInvalid, "InvalidArgument: io_id %d is not a registered output", id);
        return it->second;
    }

    // decode: MozJpegDecoder::read_frame (codecs/mozjpeg_decoder.rs:295-420) on the device, or the raw extension
    FramePtr decode(int32_t io_id, uint32_t hint_w, uint32_t hint_h, bool luma_spatial, bool luma_srgb) {
        Io& in = input(io_id);
        if (in.in_len >= kRawHeader && std::memcmp(in.in, kRawMagic, 8) == 0) {
            uint32_t hdr[4];
            std::memcpy(hdr, in.in + 8, 16);
            const uint32_t w = hdr[0], h = hdr[1], stride = hdr[2];
            if (w == 0 || h == 0 || stride < w * 4ull || (stride & 3u) || in.in_len < kRawHeader + static_cast<size_t>(h - 1) * stride + w * 4ull)
                raise(kImageMalformed, "ImageMalformed: raw BGRA container header does not match its length");
            FramePtr f = new_frame(w, h, hdr[3] != 0, 0, true);
            hip_check(hipMemcpy2D(f->d, f->stride, in.in + kRawHeader, stride, w * 4ull, h, hipMemcpyHostToDevice), "upload(raw frame)");
            decodes.push_back({io_id, w, h, "application/x-imageflow-bgra", "ifbgra"});
            return f;
        }
        if (in.in_len < 3 || in.in[0] != 0xFF || in.in[1] != 0xD8)                        // codecs/mod.rs:398-415 sniffing
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: io_id %d is neither a JPEG nor the raw BGRA extension", io_id);
        ifhip_jpeg_entropy* ent = nullptr;
        const uint8_t* files[1] = {in.in};
        const size_t lens[1] = {in.in_len};
        int rc = ifhip_jpeg_entropy_create(&ent, files, lens, 1);
        if (rc == IFHIP_METHOD_NOT_IMPLEMENTED)
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: %s (baseline sequential JPEG only; keep other files on libjpeg)", ifhip_last_error_message());
        check(rc);
        struct EntGuard { ifhip_jpeg_entropy* e; ~EntGuard() { ifhip_jpeg_entropy_destroy(e); } } eg{ent};
        uint32_t w = 0, h = 0, bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, nsub = 0, nseg = 0;
        int ncomp = 0;
        uint8_t hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1};
        check(ifhip_jpeg_entropy_info(ent, &w, &h, &ncomp, hs, vs, bw, bh, &nsub, &nseg));
        // MzDec::apply_downscaling (mozjpeg_decoder.rs:588-618): smallest i/8 (7 skipped) that still covers the hint
        int scale = 8;
        if (hint_w > 0 && hint_h > 0)
            for (int i = 1; i < 8; ++i) {
                if (i == 7) continue;
                if ((static_cast<uint64_t>(w) * i + 7) / 8 >= hint_w && (static_cast<uint64_t>(h) * i + 7) / 8 >= hint_h) { scale = i; break; }
            }
        int16_t* coef[3] = {nullptr, nullptr, nullptr};
        struct CoefGuard { int16_t** p; ~CoefGuard() { for (int i = 0; i < 3; ++i) if (p[i]) (void)hipFree(p[i]); } } cg{coef};
        for (int k = 0; k < 3; ++k)
            hip_check(hipMalloc(reinterpret_cast<void**>(&coef[k]), std::max<size_t>(1, static_cast<size_t>(bw[k]) * bh[k]) * 128), "hipMalloc(coefficients)");
        uint32_t rounds = 0;
        check(ifhip_jpeg_entropy_decode_device(ent, coef[0], coef[1], coef[2], &rounds, nullptr));
        uint16_t qt[192];
        check(ifhip_jpeg_entropy_quant_tables(ent, qt));
        uint16_t* d_qt = nullptr;
        hip_check(hipMalloc(reinterpret_cast<void**>(&d_qt), sizeof qt), "hipMalloc(qt)");
        struct QtGuard { uint16_t* p; ~QtGuard() { (void)hipFree(p); } } qg{d_qt};
        hip_check(hipMemcpy(d_qt, qt, static_cast<size_t>(ncomp) * 128, hipMemcpyHostToDevice), "upload(qt)");
        ifhip_jpeg_stage* st = nullptr;
        const bool spatial = scale < 8 && luma_spatial;
        check(ifhip_jpeg_stage_create(&st, w, h, ncomp, hs, vs, scale, spatial ? 1 : 0, spatial && luma_srgb ? 1 : 0, 1));
        struct StGuard { ifhip_jpeg_stage* s; ~StGuard() { ifhip_jpeg_stage_destroy(s); } } sg{st};
        uint32_t ow = 0, oh = 0;
        check(ifhip_jpeg_stage_output_size(st, &ow, &oh));
        FramePtr f = new_frame(ow, oh, false, 0, false);                              // alpha not meaningful (:101-123)
        check(ifhip_jpeg_idct_color_batch_device(st, coef[0], coef[1], coef[2], d_qt, 1, f->d, f->bytes(), f->stride, nullptr));
        hip_check(hipStreamSynchronize(nullptr), "decode");
        decodes.push_back({io_id, w, h, "image/jpeg", "jpg"});
        return f;
    }

    // Resample2D -> CreateCanvas + Scale2d -> DrawImageExact.render -> scale_and_render (scale_render.rs:30-320)
    FramePtr resample(const FramePtr& in, uint32_t w, uint32_t h, const JVal* hints) {
        if (w == 0 || h == 0) raise(kNodeArgumentInvalid, "InvalidNodeParams: resample_2d target size must be non-zero");
        float sharpen = 0.f;
        int down = IFHIP_FILTER_ROBIDOUX, up = IFHIP_FILTER_GINSENG, space = IFHIP_SPACE_LINEAR;     // :255-259, :276-277
        uint32_t bg = 0;
        bool when_always = false;
        if (hints && !hints->is_null()) {
            if (const JVal* s = hints->get("sharpen_percent")) if (s->t == JVal::Num) sharpen = static_cast<float>(s->n);
            down = parse_filter(hints->get("down_filter"), down);
            up = parse_filter(hints->get("up_filter"), up);
            if (const JVal* cs = hints->get("scaling_colorspace"))
                if (cs->t == JVal::Str) {
                    if (cs->s == "srgb") space = IFHIP_SPACE_SRGB;
                    else if (cs->s != "linear") raise(kInvalidJson, "InvalidJson: scaling_colorspace must be srgb or linear");
                }
            bg = parse_color(hints->get("background_color"), "resample_2d.hints.background_color");
            if (const JVal* rw = hints->get("resample_when")) when_always = rw->t == JVal::Str && rw->s == "always";
        }
        const bool matte = (bg >> 24) != 0;
        // resample_when default SizeDiffersOrSharpeningRequested: a same-size, unsharpened resample is removed from the
        // graph unless a matte has to be applied to a Bgra32 parent (scale_render.rs:38-49,68-80,113-115)
        if (!when_always && w == in->w && h == in->h && sharpen <= 0.f && !(matte && in->alpha)) return in;
        const bool downscale = w < in->w || h < in->h;                                // :255-259
        FramePtr canvas = new_frame(w, h, in->alpha, matte ? bg : 0u, true);
        const int compose = matte ? IFHIP_BLEND_WITH_MATTE : IFHIP_REPLACE_SELF;      // blend: Overwrite when bg transparent (:169-192)
        ifhip_resample_plan* plan = nullptr;
        check(ifhip_resample_plan_create(&plan, in->w, in->h, w, h, downscale ? down : up, sharpen));
        struct PlanGuard { ifhip_resample_plan* p; ~PlanGuard() { ifhip_resample_plan_destroy(p); } } pg{plan};
        check(ifhip_scale_and_render_batch_device(plan, in->d, in->bytes(), in->stride, in->alpha ? 1 : 0, 1, canvas->d, canvas->bytes(),
                                                  w, h, canvas->stride, 0, 0, space, compose, bg, nullptr, -1, nullptr));
        hip_check(hipStreamSynchronize(nullptr), "resample_2d");
        if (matte && (bg >> 24) == 255) canvas->alpha = false;                        // an opaque matte leaves no meaningful alpha
        canvas->compose = IFHIP_BLEND_WITH_SELF;                                      // :314
        return canvas;
    }

    // constrain (flow/nodes/constrain.rs:41-98 -> imageflow_riapi process_constraint): the aspect-preserving modes that
    // need neither crop nor pad; target rounding as AspectRatio::proportional (imageflow_riapi/src/sizing.rs:118-185).
    FramePtr constrain(const FramePtr& in, const JVal& p) {
        const JVal* mode = p.get("mode");
        const std::string m = mode && mode->t == JVal::Str ? mode->s : "";
        const JVal *jw = p.get("w"), *jh = p.get("h");
        const bool has_w = jw && jw->t == JVal::Num, has_h = jh && jh->t == JVal::Num;
        if (m != "within" && m != "fit" && m != "distort")
            raise(kActionNotSupported, "ActionNotSupported: constrain mode '%s' (this shim: within, fit, distort)", m.c_str());
        if (!has_w && !has_h) return in;
        double tw = has_w ? jw->n : 0, th = has_h ? jh->n : 0;
        if ((has_w && tw < 1) || (has_h && th < 1)) raise(kNodeArgumentInvalid, "InvalidNodeParams: constrain w/h must be >= 1");
        uint32_t ow, oh;
        if (m == "distort") { ow = has_w ? static_cast<uint32_t>(tw) : in->w; oh = has_h ? static_cast<uint32_t>(th) : in->h; }
        else {
            const double sx = has_w ? tw / in->w : 1e300, sy = has_h ? th / in->h : 1e300;
            double s = std::min(sx, sy);
            if (m == "within" && s >= 1.0) return in;                                 // never up-scales
            const bool basis_is_width = sx <= sy;
            if (basis_is_width) { ow = static_cast<uint32_t>(tw); oh = static_cast<uint32_t>(std::max(1.0, std::round(tw * in->h / in->w))); }
            else { oh = static_cast<uint32_t>(th); ow = static_cast<uint32_t>(std::max(1.0, std::round(th * in->w / in->h))); }
            if (has_w && has_h) { ow = std::min<uint32_t>(ow, static_cast<uint32_t>(tw)); oh = std::min<uint32_t>(oh, static_cast<uint32_t>(th)); }
        }
        const JVal* hints = p.get("hints");
        return resample(in, ow, oh, hints);
    }

    // command_string {kind: "ir4", value: "width=200&..."}: the querystring form of BASELINE config 1.  Only the sizing
    // keys that reach the hot path (width/w, height/h, mode=max default; down.colorspace) -- imageflow_riapi is out of
    // scope.  JPEG pre-shrink hint exactly as Ir4Expand::get_decode_commands (imageflow_riapi/src/ir4/mod.rs:155-210).
    FramePtr command_string(const JVal& p, FramePtr in) {
        const JVal* kind = p.get("kind");
        const JVal* value = p.get("value");
        if (!kind || kind->t != JVal::Str || kind->s != "ir4" || !value || value->t != JVal::Str)
            raise(kInvalidJson, "InvalidJson: command_string needs kind \"ir4\" and a value");
        double qw = 0, qh = 0;
        bool srgb = false;
        size_t i = 0;
        const std::string& q = value->s;
        while (i < q.size()) {
            const size_t amp = std::min(q.find('&', i), q.size());
            const std::string kv = q.substr(i, amp - i);
            i = amp + 1;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
            for (char& ch : k) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
            if (k == "width" || k == "w" || k == "maxwidth") qw = std::atof(v.c_str());
            else if (k == "height" || k == "h" || k == "maxheight") qh = std::atof(v.c_str());
            else if (k == "down.colorspace") srgb = v == "srgb";
            else if (k == "mode") { if (v != "max") raise(kActionNotSupported, "ActionNotSupported: querystring mode=%s (this shim: max)", v.c_str()); }
            else if (k == "format" || k == "quality" || k == "down.filter") {}           // encode-side / default keys
            else raise(kActionNotSupported, "ActionNotSupported: querystring key '%s'", k.c_str());
        }
        const JVal* dec = p.get("decode");
        const JVal* enc = p.get("encode");
        uint32_t src_w = 0, src_h = 0;
        if (dec && dec->t == JVal::Num) {
            Io& io = input(static_cast<int32_t>(dec->n));
            int nc = 0;
            uint8_t hs[3], vs[3];
            uint32_t bw[3], bh[3], ri = 0;
            uint16_t qt[192];
            const bool is_jpeg = io.in_len > 2 && io.in[0] == 0xFF && io.in[1] == 0xD8;
            if (is_jpeg) check(ifhip_jpeg_parse_headers(io.in, io.in_len, &src_w, &src_h, &nc, hs, vs, bw, bh, qt, &ri));
        } else if (in) { src_w = in->w; src_h = in->h; }
        else raise(kGraphInvalid, "GraphInvalid: command_string has neither a decode io nor an input frame");
        auto target = [&](uint32_t sw, uint32_t sh, uint32_t* ow, uint32_t* oh) {        // mode=max: fit inside, never up-scale
            double s = 1.0;
            if (qw > 0) s = std::min(s, qw / sw);
            if (qh > 0) s = std::min(s, qh / sh);
            const bool by_w = qw > 0 && (qh <= 0 || qw / sw <= qh / sh);
            if (s >= 1.0) { *ow = sw; *oh = sh; return; }
            if (by_w) { *ow = static_cast<uint32_t>(qw); *oh = static_cast<uint32_t>(std::max(1.0, std::round(qw * sh / sw))); }
            else { *oh = static_cast<uint32_t>(qh); *ow = static_cast<uint32_t>(std::max(1.0, std::round(qh * sw / sh))); }
        };
        if (dec && dec->t == JVal::Num) {
            uint32_t hint_w = 0, hint_h = 0;
            if (src_w) {
                uint32_t ow, oh;
                target(src_w, src_h, &ow, &oh);
                const double downscale = std::min(static_cast<double>(src_w) / ow, static_cast<double>(src_h) / ow);   // sic: `to.w` twice (:161-162)
                const double preshrink = 2.1 / downscale;
                if (preshrink < 1.0) { hint_w = static_cast<uint32_t>(std::floor(src_w * preshrink)); hint_h = static_cast<uint32_t>(std::floor(src_h * preshrink)); }
            }
            in = decode(static_cast<int32_t>(dec->n), hint_w, hint_h, !srgb, !srgb);
            if (!src_w) { src_w = in->w; src_h = in->h; }
        }
        uint32_t ow, oh;
        target(src_w, src_h, &ow, &oh);
        JVal hints;
        hints.t = JVal::Obj;
        if (srgb) { JVal cs; cs.t = JVal::Str; cs.s = "srgb"; hints.o.emplace_back("scaling_colorspace", cs); }
        FramePtr out = resample(in, ow, oh, &hints);
        if (enc && enc->t == JVal::Num) encode(out, static_cast<int32_t>(enc->n), nullptr);
        return out;
    }

    // encode: EncoderPreset::LibjpegTurbo is written as a real JPEG (codecs/mozjpeg.rs:78-160, create_classic :62-77):
    // apply_matte + forward DCT / quantisation on the device, the Huffman coder and the markers on the host
    // (csrc/jpeg_write.cpp) -- baseline, with optimize_huffman_coding libjpeg's optimal tables, with progressive its
    // standard scan script (:121-129).  The content-adaptive sampling choice
    // (evalchroma, an external crate) is not reproduced: the file uses the maximum the reference allows (:133, 4:2:0).
    // EXTENSION: every other preset writes the raw BGRA container (PNG / GIF / WebP coders are out of scope).
    void encode(const FramePtr& f, int32_t io_id, const JVal* preset) {
        Io& o = output(io_id);
        const JVal* classic = preset ? preset->get("libjpeg_turbo") : nullptr;
        if (classic) {
            auto flag = [&](const char* k) { const JVal* v = classic->get(k); return v && v->t == JVal::Bool && v->b; };
            const int write_flags = (flag("progressive") ? IFHIP_JPEG_PROGRESSIVE : 0) | (flag("optimize_huffman_coding") ? IFHIP_JPEG_OPTIMIZE_HUFFMAN : 0);
            const JVal* q = classic->get("quality");
            int quality = 75;                                                            // mozjpeg.rs:32 DEFAULT_QUALITY
            if (q && q->t == JVal::Num) quality = q->n > 100 ? 100 : (q->n < 0 ? 0 : static_cast<int>(q->n));
            const JVal* m = classic->get("matte");
            const uint32_t matte = m && !m->is_null() ? parse_color(m, "encode.preset.libjpeg_turbo.matte") : 0xFFFFFFFFu;   // :88-92
            check(ifhip_apply_matte_batch_device(f->d, f->bytes(), 1, f->w, f->h, f->stride, f->alpha ? 1 : 0, matte, nullptr));
            f->alpha = false;                                                            // :94 set_alpha_meaningful(false)
            const uint8_t hs[3] = {2, 1, 1}, vs[3] = {2, 1, 1};
            uint16_t qt2[2][64], qt3[3][64];
            ifhip_jpeg_quality_tables(quality, &qt2[0][0]);
            std::memcpy(qt3[0], qt2[0], 128); std::memcpy(qt3[1], qt2[1], 128); std::memcpy(qt3[2], qt2[1], 128);
            ifhip_jpeg_fwd_stage* st = nullptr;
            check(ifhip_jpeg_fwd_stage_create(&st, f->w, f->h, hs, vs, 1));
            std::unique_ptr<ifhip_jpeg_fwd_stage, void (*)(ifhip_jpeg_fwd_stage*)> st_guard(st, ifhip_jpeg_fwd_stage_destroy);
            uint32_t bw[3], bh[3];
            check(ifhip_jpeg_fwd_stage_block_dims(st, bw, bh));
            size_t off[4] = {0, 0, 0, 0};
            for (int c = 0; c < 3; ++c) off[c + 1] = off[c] + static_cast<size_t>(bw[c]) * bh[c] * 64u;
            int16_t* d_coef = nullptr;
            uint16_t* d_qt = nullptr;
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_coef), off[3] * 2u + 384u), "hipMalloc(coefficients)");
            std::unique_ptr<int16_t, void (*)(int16_t*)> coef_guard(d_coef, [](int16_t* p) { (void)hipFree(p); });
            d_qt = reinterpret_cast<uint16_t*>(d_coef + off[3]);
            hip_check(hipMemcpy(d_qt, qt3, 384, hipMemcpyHostToDevice), "upload(quant tables)");
            check(ifhip_jpeg_forward_batch_device(st, f->d, f->bytes(), f->stride, d_qt, 1, d_coef + off[0], d_coef + off[1], d_coef + off[2], nullptr));
            std::vector<int16_t> coef(off[3]);
            hip_check(hipMemcpy(coef.data(), d_coef, off[3] * 2u, hipMemcpyDeviceToHost), "download(coefficients)");
            size_t len = 0;
            o.owned.assign(std::max<size_t>(4096u, off[3]), 0);                          // a file is smaller than its coefficients: one pass
            int wrc = ifhip_jpeg_write(coef.data() + off[0], coef.data() + off[1], coef.data() + off[2], bw, bh, 3, hs, vs, f->w, f->h, quality, write_flags,
                                       o.owned.data(), o.owned.size(), &len);
            if (wrc != IFHIP_OK && len > o.owned.size()) {
                o.owned.assign(len, 0);
                wrc = ifhip_jpeg_write(coef.data() + off[0], coef.data() + off[1], coef.data() + off[2], bw, bh, 3, hs, vs, f->w, f->h, quality, write_flags,
                                       o.owned.data(), o.owned.size(), &len);
            }
            check(wrc);
            o.owned.resize(len);
            o.written = true;
            encodes.push_back({io_id, f->w, f->h, "image/jpeg", "jpg"});
            return;
        }
        o.owned.assign(kRawHeader + f->bytes(), 0);
        std::memcpy(o.owned.data(), kRawMagic, 8);
        const uint32_t hdr[4] = {f->w, f->h, f->stride, f->alpha ? 1u : 0u};
        std::memcpy(o.owned.data() + 8, hdr, 16);
        hip_check(hipMemcpy(o.owned.data() + kRawHeader, f->d, f->bytes(), hipMemcpyDeviceToHost), "download(frame)");
        o.written = true;
        encodes.push_back({io_id, f->w, f->h, "application/x-imageflow-bgra", "ifbgra"});
    }

    FramePtr copy_into_canvas(const FramePtr& in, const FramePtr& canvas, uint32_t fx, uint32_t fy, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
        int canvas_alpha = canvas->alpha ? 1 : 0;
        check(ifhip_copy_rect_batch_device(in->d, in->bytes(), in->w, in->h, in->stride, in->alpha ? 1 : 0, canvas->d, canvas->bytes(),
                                           canvas->w, canvas->h, canvas->stride, &canvas_alpha, fx, fy, x, y, w, h, 1, nullptr));
        canvas->alpha = canvas_alpha != 0;
        hip_check(hipStreamSynchronize(nullptr), "copy_rect");
        return canvas;
    }
    FramePtr transposed(const FramePtr& in) {
        FramePtr t = new_frame(in->h, in->w, in->alpha, 0, true);
        check(ifhip_transpose_batch_device(in->d, in->bytes(), in->w, in->h, in->stride, t->d, t->bytes(), t->w, t->h, t->stride, 1, nullptr));
        return t;
    }
    FramePtr flip(const FramePtr& in, bool vertical) {
        check(vertical ? ifhip_flip_vertical_batch_device(in->d, in->bytes(), 1, in->w, in->h, in->stride, nullptr)
                       : ifhip_flip_horizontal_batch_device(in->d, in->bytes(), 1, in->w, in->h, in->stride, nullptr));
        return in;
    }

    // one node: `in` is the frame of its (single) input edge, or null for source nodes
    FramePtr run_node(const std::string& name, const JVal& p, FramePtr in) {
        auto need_input = [&] { if (!in) raise(kGraphInvalid, "GraphInvalid: node '%s' has no input frame", name.c_str()); };
        if (name == "decode") {
            uint32_t hw = 0, hh = 0;
            bool spatial = false, gamma = false;
            if (const JVal* cmds = p.get("commands"))
                if (cmds->t == JVal::Arr)
                    for (const JVal& cmd : cmds->a)
                        if (const JVal* j = cmd.get("jpeg_downscale_hints")) {               // s::JpegIDCTDownscaleHints
                            hw = want_u32(*j, "width", "jpeg_downscale_hints"); hh = want_u32(*j, "height", "jpeg_downscale_hints");
                            if (const JVal* b = j->get("scale_luma_spatially")) spatial = b->t == JVal::Bool && b->b;
                            if (const JVal* b = j->get("gamma_correct_for_srgb_during_spatial_luma_scaling")) gamma = b->t == JVal::Bool && b->b;
                        }
            return decode(static_cast<int32_t>(want_int(p, "io_id", "decode")), hw, hh, spatial, gamma);
        }
        if (name == "create_canvas") {
            const JVal* fmt = p.get("format");
            const std::string f = fmt && fmt->t == JVal::Str ? fmt->s : "bgra_32";
            if (f != "bgra_32" && f != "bgr_32") raise(kActionNotSupported, "ActionNotSupported: create_canvas format %s", f.c_str());
            return new_frame(want_u32(p, "w", "create_canvas"), want_u32(p, "h", "create_canvas"), f == "bgra_32", parse_color(p.get("color"), "create_canvas.color"), true);
        }
        if (name == "command_string") return command_string(p, in);
        need_input();
        if (name == "resample_2d") return resample(in, want_u32(p, "w", "resample_2d"), want_u32(p, "h", "resample_2d"), p.get("hints"));
        if (name == "constrain") return constrain(in, p);
        if (name == "encode") { encode(in, static_cast<int32_t>(want_int(p, "io_id", "encode")), p.get("preset")); return in; }
        if (name == "fill_rect") {                                                    // clone_crop_fill_expand.rs:107-137
            in->compose = IFHIP_BLEND_WITH_SELF;                                      // :112: set before the fill, so matte canvases accept sub-rects
            check(ifhip_fill_rect_batch_device(in->d, in->bytes(), 1, in->w, in->h, in->stride, in->compose, want_u32(p, "x1", name.c_str()),
                                               want_u32(p, "y1", name.c_str()), want_u32(p, "x2", name.c_str()), want_u32(p, "y2", name.c_str()),
                                               parse_color(p.get("color"), "fill_rect.color"), nullptr));
            return in;
        }
        if (name == "expand_canvas") {                                                // :224-262
            const uint32_t l = want_u32(p, "left", "expand_canvas"), t = want_u32(p, "top", "expand_canvas"), r = want_u32(p, "right", "expand_canvas"),
                           b = want_u32(p, "bottom", "expand_canvas"), color = parse_color(p.get("color"), "expand_canvas.color");
            FramePtr canvas = new_frame(in->w + l + r, in->h + t + b, (color >> 24) == 255 ? in->alpha : true, color, true);
            return copy_into_canvas(in, canvas, 0, 0, in->w, in->h, l, t);
        }
        if (name == "crop") {                                                         // :519-541 (materialised: a copy)
            const uint32_t x1 = want_u32(p, "x1", "crop"), y1 = want_u32(p, "y1", "crop"), x2 = want_u32(p, "x2", "crop"), y2 = want_u32(p, "y2", "crop");
            if (x2 <= x1 || y2 <= y1 || x2 > in->w || y2 > in->h) raise(kNodeArgumentInvalid, "InvalidNodeParams: Invalid crop bounds");
            FramePtr canvas = new_frame(x2 - x1, y2 - y1, in->alpha, 0, true);
            return copy_into_canvas(in, canvas, x1, y1, x2 - x1, y2 - y1, 0, 0);
        }
        if (name == "flip_v") return flip(in, true);
        if (name == "flip_h") return flip(in, false);
        if (name == "transpose") return transposed(in);
        if (name == "rotate_90") return flip(transposed(in), false);                  // rotate_flip_transpose.rs:51-66
        if (name == "rotate_180") return flip(flip(in, true), false);
        if (name == "rotate_270") return flip(transposed(in), true);
        raise(kActionNotSupported, "ActionNotSupported: node '%s' is outside the pixel hot path this library replaces", name.c_str());
    }

    static void node_of(const JVal& n, std::string* name, const JVal** params) {
        if (n.t == JVal::Str) { *name = n.s; static const JVal kNull; *params = &kNull; return; }
        if (n.t != JVal::Obj || n.o.size() != 1) raise(kInvalidJson, "InvalidJson: a node is {\"name\": {params}}");
        *name = n.o[0].first;
        *params = &n.o[0].second;
    }

    void run_framewise(const JVal& fw) {
        if (const JVal* steps = fw.get("steps")) {
            if (steps->t != JVal::Arr) raise(kInvalidJson, "InvalidJson: framewise.steps must be an array");
            FramePtr cur;
            for (const JVal& n : steps->a) {
                std::string name;
                const JVal* params;
                node_of(n, &name, &params);
                cur = run_node(name, *params, cur);
            }
            return;
        }
        const JVal* graph = fw.get("graph");
        if (!graph) raise(kInvalidJson, "InvalidJson: framewise needs steps or graph");
        const JVal *nodes = graph->get("nodes"), *edges = graph->get("edges");
        if (!nodes || nodes->t != JVal::Obj || !edges || edges->t != JVal::Arr) raise(kInvalidJson, "InvalidJson: graph needs nodes{} and edges[]");
        std::map<int64_t, const JVal*> node_by_id;
        std::map<int64_t, int64_t> parent;
        for (const auto& kv : nodes->o) node_by_id[std::atoll(kv.first.c_str())] = &kv.second;
        for (const JVal& e : edges->a) {
            const int64_t from = want_int(e, "from", "edge"), to = want_int(e, "to", "edge");
            const JVal* kind = e.get("kind");
            if (!kind || kind->t != JVal::Str || kind->s != "input") raise(kActionNotSupported, "ActionNotSupported: only `input` edges (canvas edges are outside this shim)");
            if (!node_by_id.count(from) || !node_by_id.count(to)) raise(kGraphInvalid, "GraphInvalid: edge names a missing node");
            if (parent.count(to)) raise(kGraphInvalid, "GraphInvalid: node %lld has two input edges", static_cast<long long>(to));
            parent[to] = from;
        }
        std::map<int64_t, FramePtr> done;
        std::map<int64_t, int> state;                                                  // 1 = on the stack (cycle check)
        // nodes that mutate their input in place must not see a frame another consumer still needs: give every node
        // with a shared parent its own copy
        std::map<int64_t, int> consumers;
        for (const auto& kv : parent) ++consumers[kv.second];
        std::function<FramePtr(int64_t)> eval = [&](int64_t id) -> FramePtr {
            auto it = done.find(id);
            if (it != done.end()) return it->second;
            if (state[id] == 1) raise(kGraphInvalid, "GraphInvalid: cycle through node %lld", static_cast<long long>(id));
            state[id] = 1;
            FramePtr in;
            auto pit = parent.find(id);
            if (pit != parent.end()) {
                in = eval(pit->second);
                std::string nm;
                const JVal* pp;
                node_of(*node_by_id[id], &nm, &pp);
                const bool mutates = nm == "fill_rect" || nm == "flip_v" || nm == "flip_h" || nm == "rotate_180";
                if (in && mutates && consumers[pit->second] > 1) {
                    FramePtr c = new_frame(in->w, in->h, in->alpha, 0, false);
                    hip_check(hipMemcpy(c->d, in->d, in->bytes(), hipMemcpyDeviceToDevice), "clone");
                    c->compose = in->compose; c->matte = in->matte;
                    in = c;
                }
            }
            std::string name;
            const JVal* params;
            node_of(*node_by_id[id], &name, &params);
            FramePtr out = run_node(name, *params, in);
            state[id] = 2;
            done[id] = out;
            return out;
        };
        for (const auto& kv : node_by_id) eval(kv.first);
    }
};

// Build001.io (imageflow_types/src/lib.rs:1433-1456, 1577-1581): placeholder / output_buffer need the buffers registered
// through the ABI; bytes_hex and base_64 carry the bytes inline.
void add_io_from_json(imageflow_context* c, const JVal& ios) {
    if (ios.t != JVal::Arr) raise(kInvalidJson, "InvalidJson: io must be an array");
    for (const JVal& o : ios.a) {
        const int32_t id = static_cast<int32_t>(want_int(o, "io_id", "io"));
        const JVal* dir = o.get("direction");
        const JVal* io = o.get("io");
        if (!dir || dir->t != JVal::Str || !io) raise(kInvalidJson, "InvalidJson: io entries need direction and io");
        const bool out = dir->s == "out";
        if (io->t == JVal::Str && io->s == "placeholder") {
            if (!c->io.count(id)) raise(kArgumentInvalid, "InvalidArgument: io_id %d is a placeholder but no buffer was added for it", id);
            continue;
        }
        if (c->io.count(id)) raise(kArgumentInvalid, "InvalidArgument: io_id %d is already in use", id);
        Io e;
        e.is_output = out;
        if (io->t == JVal::Str && io->s == "output_buffer") { if (!out) raise(kInvalidJson, "InvalidJson: output_buffer on an input"); }
        else if (const JVal* hex = io->get("bytes_hex")) {
            if (out || hex->t != JVal::Str || (hex->s.size() & 1)) raise(kInvalidJson, "InvalidJson: bad bytes_hex");
            for (size_t i = 0; i < hex->s.size(); i += 2) e.owned.push_back(static_cast<uint8_t>(std::strtoul(hex->s.substr(i, 2).c_str(), nullptr, 16)));
            e.in_len = e.owned.size();
        } else if (const JVal* b64 = io->get("base_64")) {
            if (out || b64->t != JVal::Str) raise(kInvalidJson, "InvalidJson: bad base_64");
            uint32_t acc = 0;
            int bits = 0;
            for (unsigned char ch : b64->s) {
                int v = ch >= 'A' && ch <= 'Z' ? ch - 'A' : ch >= 'a' && ch <= 'z' ? ch - 'a' + 26 : ch >= '0' && ch <= '9' ? ch - '0' + 52 : ch == '+' ? 62 : ch == '/' ? 63 : -1;
                if (v < 0) continue;
                acc = (acc << 6) | static_cast<uint32_t>(v); bits += 6;
                if (bits >= 8) { bits -= 8; e.owned.push_back(static_cast<uint8_t>(acc >> bits)); }
            }
            e.in_len = e.owned.size();
        } else raise(kActionNotSupported, "ActionNotSupported: io kind (this shim: placeholder, output_buffer, bytes_hex, base_64)");
        auto& slot = c->io[id] = std::move(e);
        if (!slot.is_output) slot.in = slot.owned.data();
    }
}

std::string job_result_json(const Job& job, const char* key) {
    std::string s = "{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"" + std::string(key) + "\": {\n      \"encodes\": [";
    for (size_t i = 0; i < job.encodes.size(); ++i) {
        const EncodeRecord& e = job.encodes[i];
        s += std::string(i ? "," : "") + "\n        {\"preferred_mime_type\": \"" + e.mime + "\", \"preferred_extension\": \"" + e.ext + "\", \"io_id\": " +
             std::to_string(e.io_id) + ", \"w\": " + std::to_string(e.w) + ", \"h\": " + std::to_string(e.h) + ", \"bytes\": \"elsewhere\"}";
    }
    s += "\n      ],\n      \"decodes\": [";
    for (size_t i = 0; i < job.decodes.size(); ++i) {
        const DecodeRecord& d = job.decodes[i];
        s += std::string(i ? "," : "") + "\n        {\"preferred_mime_type\": \"" + d.mime + "\", \"preferred_extension\": \"" + d.ext + "\", \"io_id\": " +
             std::to_string(d.io_id) + ", \"w\": " + std::to_string(d.w) + ", \"h\": " + std::to_string(d.h) + "}";
    }
    s += "\n      ],\n      \"performance\": null\n    }\n  }\n}";
    return s;
}

[[noreturn]] void abort_null_context() {                          // imageflow_abi/src/lib.rs:309-325
    fprintf(stderr, "Null context pointer provided. Terminating process.\n");
    std::abort();
}
#define CTX_OR_ABORT(c) do { if (!(c)) abort_null_context(); } while (0)

}  // namespace

// ==================================================================================================================
extern "C" {

bool imageflow_abi_compatible(uint32_t major, uint32_t minor) { return major == IMAGEFLOW_ABI_VER_MAJOR && minor <= IMAGEFLOW_ABI_VER_MINOR; }
uint32_t imageflow_abi_version_major(void) { return IMAGEFLOW_ABI_VER_MAJOR; }
uint32_t imageflow_abi_version_minor(void) { return IMAGEFLOW_ABI_VER_MINOR; }

struct imageflow_context* imageflow_context_create(uint32_t major, uint32_t minor) {       // lib.rs:430
    if (!imageflow_abi_compatible(major, minor)) return nullptr;
    try { return new imageflow_context; } catch (...) { return nullptr; }
}
bool imageflow_context_begin_terminate(struct imageflow_context* c) { CTX_OR_ABORT(c); return true; }
void imageflow_context_destroy(struct imageflow_context* c) { delete c; }

bool imageflow_context_has_error(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return c->err_cat != kOk; }
int32_t imageflow_context_error_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return c->err_cat; }
int32_t imageflow_context_error_as_exit_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return exit_code(c->err_cat); }
int32_t imageflow_context_error_as_http_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return http_code(c->err_cat); }
bool imageflow_context_error_recoverable(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    return c->err_cat != kOk && c->err_cat != kOutOfMemory && c->err_cat != kInternalError;
}
bool imageflow_context_error_try_clear(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->err_cat == kOutOfMemory || c->err_cat == kInternalError) return false;
    c->err_cat = kOk;
    c->err_msg.clear();
    return true;
}
bool imageflow_context_error_write_to_buffer(struct imageflow_context* c, char* buffer, size_t buffer_length, size_t* bytes_written) {   // lib.rs:684
    CTX_OR_ABORT(c);
    if (!buffer || buffer_length == 0 || (buffer_length >> (sizeof(size_t) * 8 - 1))) { if (bytes_written) *bytes_written = 0; return false; }
    std::lock_guard<std::mutex> lk(c->mu);
    const std::string& m = c->err_msg;
    static const char kTrunc[] = "\n[truncated]\n";
    bool whole = m.size() + 1 <= buffer_length;
    size_t n;
    if (whole) { n = m.size(); std::memcpy(buffer, m.data(), n); }
    else {
        const size_t t = sizeof kTrunc - 1;
        const size_t keep = buffer_length > t + 1 ? buffer_length - 1 - t : 0;
        std::memcpy(buffer, m.data(), keep);
        const size_t tn = std::min(t, buffer_length - 1 - keep);
        std::memcpy(buffer + keep, kTrunc, tn);
        n = keep + tn;
    }
    buffer[n] = 0;
    if (bytes_written) *bytes_written = n;
    return whole;
}

bool imageflow_context_add_input_buffer(struct imageflow_context* c, int32_t io_id, const uint8_t* buffer, size_t len, imageflow_lifetime lifetime) {   // lib.rs:1137
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!buffer) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'buffer' is null."); return false; }
    if (len >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: buffer_byte_count has its leading bit set"); return false; }
    if (c->io.count(io_id)) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is already in use"); return false; }
    Io e;
    if (lifetime == imageflow_lifetime_lifetime_outlives_context) { e.in = buffer; e.in_len = len; }
    else { e.owned.assign(buffer, buffer + len); e.in_len = len; }
    auto& slot = c->io[io_id] = std::move(e);
    if (lifetime != imageflow_lifetime_lifetime_outlives_context) slot.in = slot.owned.data();
    return true;
}
bool imageflow_context_add_output_buffer(struct imageflow_context* c, int32_t io_id) {       // lib.rs:1224
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->io.count(io_id)) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is already in use"); return false; }
    Io e;
    e.is_output = true;
    c->io[io_id] = std::move(e);
    return true;
}
bool imageflow_context_get_output_buffer_by_id(struct imageflow_context* c, int32_t io_id, const uint8_t** result_buffer, size_t* result_buffer_length) {   // lib.rs:1272
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!result_buffer || !result_buffer_length) { c->set_error(kArgumentInvalid, "NullArgument: result pointers are null"); return false; }
    auto it = c->io.find(io_id);
    if (it == c->io.end() || !it->second.is_output) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is not an output buffer"); return false; }
    *result_buffer = it->second.owned.data();
    *result_buffer_length = it->second.owned.size();
    return true;
}

const struct imageflow_json_response* imageflow_context_send_json(struct imageflow_context* c, const char* method, const uint8_t* json_buffer, size_t json_buffer_size) {   // lib.rs:944
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);                         // operations serialise per context (lib.rs:13-33)
    if (!method) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'method' is null."); return nullptr; }
    if (!json_buffer) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'json_buffer' is null."); return nullptr; }
    if (json_buffer_size >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: Argument `json_buffer_size` likely came from a negative integer."); return nullptr; }
    try {
        const std::string m = method;
        if (m == "v1/get_version_info")
            return respond(c, 200, std::string("{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"version_info\": {\"long_version_string\": \"") +
                                       ifhip_version() + " (libimageflow ABI subset " + std::to_string(IMAGEFLOW_ABI_VER_MAJOR) + "." + std::to_string(IMAGEFLOW_ABI_VER_MINOR) + ")\"}\n  }\n}");
        const bool build = m == "v1/build" || m == "v0.1/build", execute = m == "v1/execute" || m == "v0.1/execute";
        if (!build && !execute && m != "v1/get_image_info" && m != "v0.1/get_image_info") {
            c->set_error(kArgumentInvalid, "InvalidMessageEndpoint: " + m);
            return respond(c, 404, "{\n  \"success\": \"false\",\n  \"code\": 404,\n  \"message\": \"Endpoint name not understood\"}");   // json/mod.rs:158-168
        }
        const JVal root = parse_json(json_buffer, json_buffer_size);
        if (root.t != JVal::Obj) raise(kInvalidJson, "InvalidJson: the message must be an object");
        Job job{c, {}, {}};
        if (!build && !execute) {                                    // get_image_info {io_id}: header facts only
            Io& in = job.input(static_cast<int32_t>(want_int(root, "io_id", "get_image_info")));
            uint32_t w = 0, h = 0, bw[3], bh[3], ri = 0;
            int nc = 0;
            uint8_t hs[3], vs[3];
            uint16_t qt[192];
            check(ifhip_jpeg_parse_headers(in.in, in.in_len, &w, &h, &nc, hs, vs, bw, bh, qt, &ri));
            return respond(c, 200, "{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"image_info\": {\"preferred_mime_type\": \"image/jpeg\", "
                                   "\"preferred_extension\": \"jpg\", \"image_width\": " + std::to_string(w) + ", \"image_height\": " + std::to_string(h) +
                                   ", \"frame_decodes_into\": \"bgr_32\"}\n  }\n}");
        }
        if (build) if (const JVal* ios = root.get("io")) add_io_from_json(c, *ios);
        const JVal* fw = root.get("framewise");
        if (!fw || fw->t != JVal::Obj) raise(kInvalidJson, "InvalidJson: missing framewise");
        job.run_framewise(*fw);
        return respond(c, 200, job_result_json(job, build ? "build_result" : "job_result"));
    } catch (const FlowErr& e) {
        return respond_error(c, e.cat, e.msg);
    } catch (const std::bad_alloc&) {
        return respond_error(c, kOutOfMemory, "AllocationFailed: host memory");
    } catch (const std::exception& e) {                              // the catch_unwind of lib.rs:973-1017
        c->set_error(kInternalError, std::string("InternalError: ") + e.what());
        return nullptr;
    }
}

bool imageflow_json_response_read(struct imageflow_context* c, const struct imageflow_json_response* r, int64_t* status, const uint8_t** buf, size_t* len) {   // lib.rs:783
    CTX_OR_ABORT(c);
    if (!r) { std::lock_guard<std::mutex> lk(c->mu); c->set_error(kArgumentInvalid, "NullArgument: The argument response_in is null."); return false; }
    if (status) *status = r->r.status;
    if (buf) *buf = reinterpret_cast<const uint8_t*>(r->r.json.data());
    if (len) *len = r->r.json.size();
    return true;
}
bool imageflow_json_response_destroy(struct imageflow_context* c, struct imageflow_json_response* r) {   // lib.rs:842
    CTX_OR_ABORT(c);
    if (!r) return true;
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->responses.begin(); it != c->responses.end(); ++it)
        if (it->get() == r) { c->responses.erase(it); return true; }
    return false;
}

void* imageflow_context_memory_allocate(struct imageflow_context* c, size_t bytes, const char* /*filename*/, int32_t /*line*/) {   // lib.rs:1424
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (bytes >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: bytes has its leading bit set"); return nullptr; }
    try {
        c->allocations.emplace_back(new uint8_t[bytes ? bytes : 1]());
        return c->allocations.back().get();
    } catch (...) { c->set_error(kOutOfMemory, "AllocationFailed"); return nullptr; }
}
bool imageflow_context_memory_free(struct imageflow_context* c, void* p, const char* /*filename*/, int32_t /*line*/) {   // lib.rs:1484
    CTX_OR_ABORT(c);
    if (!p) return true;
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->allocations.begin(); it != c->allocations.end(); ++it)
        if (it->get() == p) { c->allocations.erase(it); return true; }
    return false;
}

}  // extern "C"
