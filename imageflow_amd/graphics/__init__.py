"""Mirror of imageflow_core::graphics for the resample / flatten path."""
