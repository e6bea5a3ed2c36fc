"""scale_render (DrawImageExact), color (ColorFilterSrgb), rotate_flip_transpose, clone_crop_fill_expand, watermark."""
