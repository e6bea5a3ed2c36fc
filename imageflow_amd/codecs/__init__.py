"""Mirror of imageflow_core::codecs for the JPEG stages: mozjpeg_decoder (entropy + pixel stage, block scalers),
mozjpeg (encode-side pixel stage)."""
