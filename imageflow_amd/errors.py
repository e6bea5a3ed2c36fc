"""FlowError / ErrorKind subset raised on this path (imageflow_core/src/errors.rs:158-248)."""
import enum


class ErrorKind(enum.IntEnum):
    Ok = 0
    InvalidArgument = 1
    MethodNotImplemented = 2
    InvalidState = 3
    AllocationFailed = 4
    GpuUnavailable = 5
    GpuError = 6


class FlowError(RuntimeError):
    def __init__(self, kind, message=""):
        self.kind = ErrorKind(kind)
        super().__init__(message or self.kind.name)
