// Constraint layout (see layout.hpp).  A restatement of the parts of imageflow_riapi that `process_constraint` runs for
// the nine ConstraintMode values: AspectRatio (sizing.rs:14-222), Layout and its steps (sizing.rs:267-471), the step
// programs per (mode, scale) pair (ir4/layout.rs:160-283), target size (:105-139), gravity (:673-698), results (:334-412).
// Integer and f64 / f32 arithmetic in the reference's order, so that a size that rounds at .5 rounds the same way.
#include "layout.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace ifhip {
namespace {

struct LayoutErr { std::string text; };

struct AR {                                            // sizing::AspectRatio: w, h >= 1
    int32_t w, h;
    double ratio() const { return static_cast<double>(w) / static_cast<double>(h); }          // :44-46
    bool aspect_wider_than(const AR& other) const { return other.ratio() > ratio(); }         // :54-56
    bool exceeds_any(const AR& o) const { return w > o.w || h > o.h; }                         // :200-202
};
std::string dbg(const AR& a) { return std::to_string(a.w) + "x" + std::to_string(a.h); }

AR create(int64_t w, int64_t h) {                      // AspectRatio::create (:35-42)
    if (w < 1 || h < 1) throw LayoutErr{"InvalidDimensions { w: " + std::to_string(w) + ", h: " + std::to_string(h) + " }"};
    return AR{static_cast<int32_t>(w), static_cast<int32_t>(h)};
}

enum BoxKind { kInner, kOuter };

// AspectRatio::proportional (:118-181) with rounding_loss_based_on_target_width / _height (:81-115)
int32_t proportional(const AR& self, int32_t basis, bool basis_is_width, const AR* target) {
    double snap_amount = 1.0 - 2.220446049250313e-16;                                          // 1f64 - f64::EPSILON
    const double ratio = self.ratio();
    if (target) {
        if (!basis_is_width) {
            const double target_x_to_self_x = static_cast<double>(target->w) / static_cast<double>(self.w);
            const double rounded_y = std::round(static_cast<double>(self.h) * target_x_to_self_x);
            snap_amount = std::fabs(static_cast<double>(target->w) - rounded_y * ratio);
        } else {
            const double target_y_to_self_y = static_cast<double>(target->h) / static_cast<double>(self.h);
            const double rounded_x = std::round(static_cast<double>(self.w) * target_y_to_self_y);
            snap_amount = std::fabs(static_cast<double>(target->h) - rounded_x / ratio);
        }
    }
    const int32_t snap_a = basis_is_width ? self.h : self.w;
    const int32_t snap_b = target ? (basis_is_width ? target->h : target->w) : snap_a;
    const double f = basis_is_width ? static_cast<double>(basis) / ratio : ratio * static_cast<double>(basis);
    const double delta_a = f - static_cast<double>(snap_a), delta_b = f - static_cast<double>(snap_b);
    int64_t v;
    if (std::fabs(delta_a) <= snap_amount && std::fabs(delta_a) <= std::fabs(delta_b)) v = snap_a;
    else if (std::fabs(delta_b) <= snap_amount) v = snap_b;
    else {
        const double rounded = std::round(f);
        if (rounded <= -2147483648.0 || rounded >= 2147483647.0) throw LayoutErr{"ValueScalingFailed"};
        v = static_cast<int64_t>(rounded);
    }
    if (v < 0) throw LayoutErr{"ValueScalingFailed"};
    return v == 0 ? 1 : static_cast<int32_t>(v);
}

AR box_of(const AR& self, const AR& target, BoxKind kind) {                                   // :185-193
    if (target.aspect_wider_than(self) == (kind == kInner)) return create(target.w, proportional(self, target.w, true, &target));
    return create(proportional(self, target.h, false, &target), target.h);
}
AR intersection(const AR& a, const AR& b) { return create(std::min(a.w, b.w), std::min(a.h, b.h)); }   // :207-209
AR distort_with(const AR& self, const AR& other_old, const AR& other_new) {                    // :211-219, mult_fraction :245-247
    return create(static_cast<int32_t>(static_cast<int64_t>(self.w) * other_new.w / other_old.w),
                  static_cast<int32_t>(static_cast<int64_t>(self.h) * other_new.h / other_old.h));
}

struct Layout {                                        // sizing::Layout (:267-274)
    AR source, target, canvas, image;
    void scale_canvas(BoxKind kind) {                  // :304-311 (target = self.target)
        const AR nc = box_of(canvas, target, kind);
        image = distort_with(image, canvas, nc);
        canvas = nc;
    }
    void distort_canvas(const AR& t) { image = distort_with(image, canvas, t); canvas = t; }   // :318-325
    void pad_canvas(const AR& t) {                     // :332-337
        if (canvas.exceeds_any(t)) throw LayoutErr{"ImpossiblePad { target: " + dbg(t) + ", current: " + dbg(canvas) + " }"};
        canvas = t;
    }
    void crop(const AR& t) {                           // :339-346
        if (t.exceeds_any(canvas)) throw LayoutErr{"ImpossibleCrop { target: " + dbg(t) + ", current: " + dbg(canvas) + " }"};
        const AR ni = intersection(image, t);
        source = box_of(ni, source, kInner);
        image = ni;
        canvas = t;
    }
    int cmp_w() const { return canvas.w < target.w ? -1 : canvas.w > target.w ? 1 : 0; }      // canvas.cmp_size(&target) (:221-223, :419-421)
    int cmp_h() const { return canvas.h < target.h ? -1 : canvas.h > target.h ? 1 : 0; }
    bool either(int o) const { return cmp_w() == o || cmp_h() == o; }                          // Cond::Either / Neither (:538-540)
    bool neither(int o) const { return cmp_w() != o && cmp_h() != o; }
    bool larger_1d_smaller_1d() const { return (cmp_w() > 0 && cmp_h() < 0) || (cmp_w() < 0 && cmp_h() > 0); }   // :524-527
};

// gravity1d (ir4/layout.rs:673-683)
int32_t gravity1d(float align_percentage, int32_t inner, int32_t outer) {
    const float ratio = std::min(std::max(align_percentage, 0.f), 100.f) / 100.f;
    if ((outer < inner && inner < 1) || outer < 1) throw LayoutErr{"Outer box should never be smaller than inner box. All values must > 0"};
    const float v = std::round(static_cast<float>(outer - inner) * ratio);
    return std::max<int32_t>(0, std::min<int32_t>(static_cast<int32_t>(v), outer - inner));
}

}  // namespace

int constraint_mode_from_name(const std::string& n) {
    static const char* const names[] = {"distort", "within", "fit", "larger_than", "within_crop", "fit_crop", "aspect_crop", "within_pad", "fit_pad"};
    for (int i = 0; i < 9; ++i)
        if (n == names[i]) return i;
    return -1;
}

bool process_constraint(int mode, int32_t source_w, int32_t source_h, int64_t w, int64_t h, bool has_gravity, float gx, float gy,
                        ConstraintLayout* out, std::string* error) {
    try {
        const AR initial = create(source_w, source_h);
        // get_wh_from_all (:62-91) without the legacy max values; get_ideal_target_size (:93-139) at zoom 1, pre-shrink ratio 1
        const bool some_w = w >= 1 && w <= 2147483647, some_h = h >= 1 && h <= 2147483647;
        AR target = initial;
        if (some_w && some_h) target = create(w, h);
        else if (some_w) target = create(w, proportional(initial, static_cast<int32_t>(w), true, nullptr));
        else if (some_h) target = create(proportional(initial, static_cast<int32_t>(h), false, nullptr), h);
        // build_constraints (:160-283): both sides ABSENT (not merely < 1) forces FitMode::Max
        enum Fit { kMax, kPad, kStretch, kCrop, kAspect } fit;
        enum Scale { kDown, kUp, kBoth } scale = kDown;
        switch (mode) {                                                                        // get_instructions (:290-332)
        case kDistort: fit = kStretch; scale = kBoth; break;
        case kWithin: fit = kMax; scale = kDown; break;
        case kFit: fit = kMax; scale = kBoth; break;
        case kLargerThan: fit = kMax; scale = kUp; break;
        case kWithinCrop: fit = kCrop; scale = kDown; break;
        case kFitCrop: fit = kCrop; scale = kBoth; break;
        case kAspectCrop: fit = kAspect; scale = kDown; break;
        case kWithinPad: fit = kPad; scale = kDown; break;
        case kFitPad: fit = kPad; scale = kBoth; break;
        default: throw LayoutErr{"NotImplemented"};
        }
        if (w < 0 && h < 0) fit = kMax;
        Layout lay{initial, target, initial, initial};                                         // Layout::create (:452-454)
        // the step programs, run as execute_all runs them (:422-450): a failed SkipUnless / a met SkipIf skips to the next
        // BeginSequence; conditions compare the CURRENT canvas with the target
        const bool gate_down = lay.either(1), gate_up = lay.neither(1);                        // Either(Greater) / Neither(Greater)
        if (fit == kMax) {
            if (scale == kBoth || (scale == kDown && gate_down) || (scale == kUp && gate_up)) lay.scale_canvas(kInner);
        } else if (fit == kPad) {
            if (scale == kBoth || (scale == kDown && gate_down) || (scale == kUp && gate_up)) { lay.scale_canvas(kInner); lay.pad_canvas(lay.target); }
        } else if (fit == kStretch) {
            if (scale == kBoth || (scale == kDown && gate_down) || (scale == kUp && gate_up)) lay.distort_canvas(lay.target);
        } else if (fit == kCrop) {
            if (scale == kBoth || (scale == kUp && gate_up)) { lay.scale_canvas(kOuter); lay.crop(lay.target); }
            else if (scale == kDown) {
                if (!lay.either(-1)) { lay.scale_canvas(kOuter); lay.crop(lay.target); }        // skip_if(Either(Less))
                if (lay.larger_1d_smaller_1d()) lay.crop(intersection(lay.image, lay.target));  // new_seq().skip_unless(Larger1DSmaller1D).crop_intersection()
            }
        } else {
            lay.crop(box_of(lay.target, lay.canvas, kInner));                                   // CropAspect (:403)
        }
        // results (:359-411)
        const float x = has_gravity ? gx : 50.f, y = has_gravity ? gy : 50.f;
        const AR new_crop = lay.source;
        const int32_t cx1 = gravity1d(x, new_crop.w, initial.w), cy1 = gravity1d(y, new_crop.h, initial.h);
        ConstraintLayout r;
        if (cx1 > 0 || cy1 > 0 || initial.w != new_crop.w || initial.h != new_crop.h) {
            r.has_crop = true;
            r.crop[0] = static_cast<uint32_t>(cx1); r.crop[1] = static_cast<uint32_t>(cy1);
            r.crop[2] = static_cast<uint32_t>(cx1 + new_crop.w); r.crop[3] = static_cast<uint32_t>(cy1 + new_crop.h);
        }
        r.scale_w = lay.image.w; r.scale_h = lay.image.h;
        r.canvas_w = lay.canvas.w; r.canvas_h = lay.canvas.h;
        const int32_t left = gravity1d(x, lay.image.w, lay.canvas.w), top = gravity1d(y, lay.image.h, lay.canvas.h);
        const int32_t right = lay.canvas.w - lay.image.w - left, bottom = lay.canvas.h - lay.image.h - top;
        if (left > 0 || top > 0 || right > 0 || bottom > 0) {
            if (left < 0 || top < 0 || right < 0 || bottom < 0) throw LayoutErr{"Negative padding showed up"};
            r.has_pad = true;
            r.pad[0] = static_cast<uint32_t>(left); r.pad[1] = static_cast<uint32_t>(top);
            r.pad[2] = static_cast<uint32_t>(right); r.pad[3] = static_cast<uint32_t>(bottom);
        }
        *out = r;
        return true;
    } catch (const LayoutErr& e) {
        if (error) *error = e.text;
        return false;
    }
}

}  // namespace ifhip
