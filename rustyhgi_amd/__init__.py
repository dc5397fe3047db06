"""rustyhgi_amd -- MI355X-native (gfx950) HGI encode/decode core behind the surface of pl0q1n/RustyHGI.

Crate-root re-exports as in the reference's src/lib.rs:16-23:
    rustyhgi_amd.{Archive, Metadata, Decoder, Encoder}, rustyhgi_amd.interpolator, rustyhgi_amd.quantizator
plus rustyhgi_amd.entropy (device-side byte histogram of the grid) and rustyhgi_amd.Planes (device
buffers placed for MI355X's HBM regions, include/hgi.h hgi_planes_alloc).
All computation happens in libhgi_hip.so (hand-written HIP kernels); see include/hgi.h.
"""
from . import entropy, interpolator, quantizator
from ._ffi import Context, HgiError, default_context
from .archive import Archive, Metadata
from .codec import Decoder, Encoder
from .grid import Grid
from .planes import Planes

__all__ = ["Encoder", "Decoder", "Grid", "Archive", "Metadata", "Context", "HgiError", "default_context", "interpolator",
           "quantizator", "entropy", "Planes"]
