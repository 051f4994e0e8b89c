"""MI355X-native backend for the hot path of the lizardxcc/ray_tracer spectral path tracer.

The product is ``lib/libmi355rt.so`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/mi355rt.h``).  This package is the thin host side used by the tests and the benchmark:
ctypes bindings (`capi`), the reference's file formats (`scene_io`), its camera (`camera`), a
mirror of its ``Renderer`` interface (`renderer`) and the synthetic workloads (`synth`).
There is no CPU fallback anywhere: importing `capi` without the built library raises.
"""
__all__ = ["capi", "scene_io", "camera", "renderer", "synth"]
