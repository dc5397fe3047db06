"""Entropy front end on the device (the reference has no counterpart -- it hands the grid to DEFLATE
on the CPU, src/archive.rs:36).  A per-frame byte histogram of the residual grid is what an entropy coder, or a
rate estimate for choosing the quantization level, starts from; it is computed where the grid already lives.
"""
import numpy as np

from . import _ffi
from .codec import _torch_ctx


def histogram(grids, context=None):
    """(B, H, W) uint8 CUDA tensor (or (H, W)) -> (B, 256) int64 CUDA tensor: counts of each residual value per frame.
    Asynchronous on the current stream, like the codec's batch calls."""
    import torch
    if grids.dim() == 2:
        grids = grids.unsqueeze(0)
    if grids.dtype != torch.uint8 or not grids.is_cuda or not grids.is_contiguous():
        raise TypeError("histogram() takes a contiguous uint8 CUDA tensor of shape (B, H, W)")
    ctx = _torch_ctx(grids, context)
    b, h, w = grids.shape
    hist = torch.empty((b, 256), dtype=torch.int64, device=grids.device)
    _ffi.check(_ffi.lib().hgi_histogram_u8_dev(ctx.handle, grids.data_ptr(), w, h, b, h * w, hist.data_ptr()))
    return hist


def entropy_bits_per_pixel(hist):
    """Order-0 entropy of each frame from its histogram: the bits per pixel an ideal memoryless coder would spend."""
    h = np.asarray(hist.cpu() if hasattr(hist, "cpu") else hist, dtype=np.float64)
    if h.ndim == 1:
        h = h[None]
    n = h.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(n > 0, h / n, 0.0)
        e = -(np.where(p > 0, p * np.log2(p), 0.0)).sum(axis=1)
    return e


def estimated_bytes(hist):
    """ceil(pixels * entropy / 8) per frame: what the grid would take under an ideal order-0 coder."""
    h = np.asarray(hist.cpu() if hasattr(hist, "cpu") else hist, dtype=np.float64)
    if h.ndim == 1:
        h = h[None]
    return np.ceil(h.sum(axis=1) * entropy_bits_per_pixel(h) / 8.0).astype(np.int64)


def deflate_grid(grid, context=None):
    """(H, W) uint8 CUDA tensor -> bytes: raw DEFLATE (one dynamic-Huffman block of literals and distance-1 run matches) of the grid's bincode image
    (u64 H*W, the bytes, u64 W) -- what follows the metadata in a .hgi archive -- entropy-coded on the device
    (hgi_deflate_grid_dev, include/hgi.h).  Synchronous."""
    import ctypes
    import torch
    if grid.dim() != 2 or grid.dtype != torch.uint8 or not grid.is_cuda or not grid.is_contiguous():
        raise TypeError("deflate_grid() takes a contiguous uint8 CUDA tensor of shape (H, W)")
    ctx = _torch_ctx(grid, context)
    h, w = grid.shape
    cap = h * w + h * w // 8 + 1024
    out = np.empty(cap, np.uint8)
    n = ctypes.c_size_t(0)
    _ffi.check(_ffi.lib().hgi_deflate_grid_dev(ctx.handle, grid.data_ptr(), w, h, out.ctypes.data, cap, ctypes.byref(n)))
    return out[:n.value].tobytes()


def deflate_grids(grids, context=None):
    """(B, H, W) uint8 CUDA tensor -> list of B bytes objects, each the raw DEFLATE stream of that grid's bincode image
    (hgi_deflate_grids_dev: the batch is pipelined in groups of frames -- one launch per phase per group, codes built on
    the host while the device histograms the next group, streams downloaded while the next group is coded)."""
    import ctypes
    import torch
    if grids.dim() != 3 or grids.dtype != torch.uint8 or not grids.is_cuda or not grids.is_contiguous():
        raise TypeError("deflate_grids() takes a contiguous uint8 CUDA tensor of shape (B, H, W)")
    ctx = _torch_ctx(grids, context)
    b, h, w = grids.shape
    cap = h * w + h * w // 8 + 1024
    out = np.empty((b, cap), np.uint8)
    sizes = (ctypes.c_size_t * b)()
    _ffi.check(_ffi.lib().hgi_deflate_grids_dev(ctx.handle, grids.data_ptr(), w, h, b, h * w, out.ctypes.data, cap, sizes))
    return [out[f, :sizes[f]].tobytes() for f in range(b)]


def deflate_grids_packed(grids, context=None, out=None):
    """As deflate_grids, through hgi_deflate_grids_packed_dev: the streams land back to back (64-byte aligned) in ONE
    host buffer and a group of frames comes down with one copy instead of one per frame.  Returns (buffer, offsets,
    sizes); stream f is buffer[offsets[f] : offsets[f] + sizes[f]].  `out`: a uint8 host array to pack into (e.g. a
    pinned torch tensor's numpy view); by default one of the worst-case size is allocated."""
    import ctypes
    import torch
    if grids.dim() != 3 or grids.dtype != torch.uint8 or not grids.is_cuda or not grids.is_contiguous():
        raise TypeError("deflate_grids_packed() takes a contiguous uint8 CUDA tensor of shape (B, H, W)")
    ctx = _torch_ctx(grids, context)
    b, h, w = grids.shape
    if out is None:
        out = np.empty(b * (h * w + h * w // 8 + 1088), np.uint8)
    sizes, offsets = (ctypes.c_size_t * b)(), (ctypes.c_size_t * b)()
    _ffi.check(_ffi.lib().hgi_deflate_grids_packed_dev(ctx.handle, grids.data_ptr(), w, h, b, h * w, out.ctypes.data, out.size, offsets, sizes))
    return out, list(offsets), list(sizes)
