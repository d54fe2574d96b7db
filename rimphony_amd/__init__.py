"""rimphony_amd -- MI355X-native batched synchrotron-coefficient integrator.

Drop-in for the per-parameter-point hot path of pkgw/rimphony (Symphony j/alpha
for Stokes I, Q, V; Heyvaerts rho_Q, rho_V) behind the C ABI of
include/rimphony_hip.h.  `api` mirrors the reference's Rust API surface; `capi`
is the raw ctypes binding; `workload` generates the synthetic benchmark batches.
Importing this package does not require a GPU; creating a Context does.
"""
from . import capi, workload  # noqa: F401

__all__ = ["capi", "workload"]
