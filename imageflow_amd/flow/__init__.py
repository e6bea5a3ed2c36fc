"""Mirror of the imageflow_core::flow node semantics that decide what reaches the kernels."""
