"""Mirror of imageflow_core::graphics: scaling, weights, color, blend, bitmaps and the whole-bitmap primitives (bitmap_ops)."""
