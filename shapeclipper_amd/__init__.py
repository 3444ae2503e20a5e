"""shapeclipper_amd -- MI355X-native (gfx950) hot path of ShapeClipper.

Layout
------
csrc/      hand-written HIP kernels + the C ABI (include/shapeclipper_hip.h)
lib/       libshapeclipper_hip.so (built in-tree by csrc/Makefile or __graft_entry__.build())
_lib.py    ctypes binding of the C ABI (raises if the library is missing -- no CPU fallback)
packing.py host-side weight packing (torch, differentiable): PE slot order, latent folding
model/ utils/   host-side mirror of the reference's call surface (model.renderer.Renderer, ...)
"""
__version__ = "0.1.0"
