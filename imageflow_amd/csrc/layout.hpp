// Constraint layout: what `constrain` and `watermark` ask imageflow_riapi for before they reach the hot path
// (flow/nodes/constrain.rs:49-52, flow/nodes/watermark.rs:134-139 -> imageflow_riapi::ir4::process_constraint).
// Host arithmetic only; restated from imageflow_riapi/src/sizing.rs and src/ir4/layout.rs (lines cited in layout.cpp).
#pragma once
#include <cstdint>
#include <string>

namespace ifhip {

enum ConstraintMode : int {            // imageflow_types ConstraintMode (lib.rs:984-1017), in its order
    kDistort = 0, kWithin, kFit, kLargerThan, kWithinCrop, kFitCrop, kAspectCrop, kWithinPad, kFitPad
};
// "distort" ... "fit_pad" -> the value above, -1 for anything else
int constraint_mode_from_name(const std::string& name);

struct ConstraintLayout {              // ir4/layout.rs ConstraintResults (:22-27)
    bool has_crop = false;
    uint32_t crop[4] = {0, 0, 0, 0};   // x1, y1, x2, y2 in the source
    int32_t scale_w = 0, scale_h = 0;  // scale_to
    bool has_pad = false;
    uint32_t pad[4] = {0, 0, 0, 0};    // left, top, right, bottom
    int32_t canvas_w = 0, canvas_h = 0;
};

// Ir4Layout::process_constraint (ir4/layout.rs:334-412).  w / h < 0: the constraint does not give that side.
// has_gravity false: ConstraintGravity::Center.  Returns false and fills *error (the Debug text of the LayoutError) when the
// reference returns Err.
bool process_constraint(int mode, int32_t source_w, int32_t source_h, int64_t w, int64_t h, bool has_gravity, float gravity_x,
                        float gravity_y, ConstraintLayout* out, std::string* error);

}  // namespace ifhip
