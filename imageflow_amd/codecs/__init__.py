"""Mirror of imageflow_core::codecs for the JPEG pixel stage."""
