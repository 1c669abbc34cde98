"""MI355X-native implementation of the PPEA-Depth training hot path.

Mirrors the reference's module surface (`ppeadepth.layers`, `ppeadepth.networks`,
`ppeadepth.trainer.Trainer.process_batch`) over hand-written gfx950 HIP kernels
(libppea_depth.so, C ABI in include/ppea_depth.h).  No CPU fallback exists.
"""
__all__ = ["layers", "networks", "ops", "trainer", "options", "dist"]
