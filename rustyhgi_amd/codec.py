"""Mirror of `hgi::{Encoder, Decoder}` (reference src/encoder.rs, src/decoder.rs) over the C ABI.

numpy arrays take the host-pointer entry points (hgi_encode_u8 / hgi_decode_u8); torch CUDA
tensors take the device-pointer, asynchronous, batched entry points on torch's current stream.
Every path ends in the HIP kernels; there is no host implementation.
"""
import ctypes

import numpy as np

from . import _ffi
from .grid import Grid
from .interpolator import Interpolator
from .quantizator import Quantizator


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _interp_id(interpolator):
    kid = getattr(interpolator, "kernel_id", None)
    if kid is None:
        raise _ffi.HgiError(_ffi.EUNSUPPORTED,
                            "interpolator %r has no device predictor" % type(interpolator).__name__)
    return int(kid)


def _torch_ctx(t, ctx=None):
    """The context serving tensor `t`, bound to torch's current stream on t's device so that the
    launches are ordered with the torch ops around them."""
    import torch
    if not t.is_cuda:
        raise ValueError("torch tensors must live on the GPU (use numpy for host buffers)")
    if t.dtype != torch.uint8 or not t.is_contiguous():
        raise ValueError("expected a contiguous uint8 tensor")
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if ctx is None:
        ctx = _ffi.default_context(dev)
    elif ctx.device != dev:
        raise ValueError("tensor lives on cuda:%d but the context was created for cuda:%d" % (dev, ctx.device))
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    return ctx


def _check_out(out, like, what):
    """A caller-supplied output buffer goes to the C ABI as a raw pointer: it must be exactly what the call will write --
    uint8, C-contiguous, the input's shape, on the input's device -- and must not share memory with the input."""
    if _is_torch(like):
        import torch
        if not _is_torch(out) or out.dtype != torch.uint8 or not out.is_contiguous() or out.shape != like.shape \
                or out.device != like.device:
            raise ValueError("%s: `out` must be a contiguous uint8 tensor of shape %s on %s" % (what, tuple(like.shape), like.device))
        a0, b0, n = like.data_ptr(), out.data_ptr(), like.numel()
    else:
        if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] \
                or not out.flags["WRITEABLE"] or out.shape != like.shape:
            raise ValueError("%s: `out` must be a writable C-contiguous uint8 array of shape %s" % (what, like.shape))
        a0, b0, n = like.ctypes.data, out.ctypes.data, like.size
    if a0 < b0 + n and b0 < a0 + n:
        raise ValueError("%s: `out` overlaps the input" % what)
    return out


def _np_image(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a (height, width) uint8 image")
    return a


class Encoder:
    """`Encoder::new(interpolator, quantizator, scale_level)` -- src/encoder.rs:18."""

    def __init__(self, interpolator, quantizator, scale_level, context=None):
        if not isinstance(interpolator, Interpolator) or not isinstance(quantizator, Quantizator):
            raise TypeError("Encoder(interpolator: Interpolator, quantizator: Quantizator, scale_level)")
        self.interpolator, self.quantizator = interpolator, quantizator
        self._interp = _interp_id(interpolator)
        self.scale_level = int(scale_level)
        self._lut = np.ascontiguousarray(quantizator.table(), dtype=np.uint8)
        self._ctx = context

    def encode(self, image):
        """`encode(GrayImage) -> Grid` -- src/encoder.rs:39.  The input is not modified."""
        if _is_torch(image):
            if image.dim() != 2:
                raise ValueError("expected a (height, width) image; use encode_batch for stacks")
            out = self.encode_batch(image.unsqueeze(0))
            return Grid(out[0], image.shape[1])
        img = _np_image(image)
        h, w = img.shape
        grid = np.empty_like(img)
        ctx = self._ctx or _ffi.default_context(0)
        _ffi.check(_ffi.lib().hgi_encode_u8(ctx.handle, img.ctypes.data, w, h, self.scale_level,
                                            self._interp, self._lut.ctypes.data,
                                            grid.ctypes.data))
        return Grid(grid, w)

    def encode_batch(self, images, out=None):
        """(B, H, W) uint8 CUDA tensor -> (B, H, W) residual planes, asynchronous on the current stream.
        A (B, H, W) numpy array (host memory) goes through hgi_encode_u8_batch instead: synchronous, the frames
        pipelined through the device so that uploads overlap downloads."""
        if not _is_torch(images):
            imgs = np.ascontiguousarray(images, dtype=np.uint8)
            if imgs.ndim != 3:
                raise ValueError("expected a (batch, height, width) stack")
            b, h, w = imgs.shape
            out = np.empty_like(imgs) if out is None else _check_out(out, imgs, "encode_batch")
            ctx = self._ctx or _ffi.default_context(0)
            _ffi.check(_ffi.lib().hgi_encode_u8_batch(ctx.handle, imgs.ctypes.data, w, h, self.scale_level, self._interp,
                                                      self._lut.ctypes.data, out.ctypes.data, b, h * w))
            return out
        import torch
        ctx = _torch_ctx(images, self._ctx)
        if images.dim() != 3:
            raise ValueError("expected a (batch, height, width) stack")
        b, h, w = images.shape
        out = torch.empty_like(images) if out is None else _check_out(out, images, "encode_batch")
        _ffi.check(_ffi.lib().hgi_encode_u8_dev(ctx.handle, images.data_ptr(), w, h, self.scale_level,
                                                self._interp, self._lut.ctypes.data,
                                                out.data_ptr(), b, h * w))
        return out


class Decoder:
    """`Decoder::new(interpolator)` -- src/decoder.rs:14."""

    def __init__(self, interpolator, context=None):
        if not isinstance(interpolator, Interpolator):
            raise TypeError("Decoder(interpolator: Interpolator)")
        self.interpolator = interpolator
        self._interp = _interp_id(interpolator)
        self._ctx = context

    def decode(self, dimensions, levels, grid):
        """`decode((width, height), levels, &Grid) -> GrayImage` -- src/decoder.rs:18."""
        width, height = int(dimensions[0]), int(dimensions[1])
        buf = grid.buffer if isinstance(grid, Grid) else grid
        if _is_torch(buf):
            return self.decode_batch(buf.reshape(1, height, width), levels)[0]
        g = np.ascontiguousarray(buf, dtype=np.uint8).reshape(height, width)
        img = np.empty_like(g)
        ctx = self._ctx or _ffi.default_context(0)
        _ffi.check(_ffi.lib().hgi_decode_u8(ctx.handle, g.ctypes.data, width, height, int(levels),
                                            self._interp, img.ctypes.data))
        return img

    def decode_batch(self, grids, levels, out=None):
        """CUDA tensor: asynchronous on the current stream; numpy stack: hgi_decode_u8_batch (see Encoder.encode_batch)."""
        if not _is_torch(grids):
            g = np.ascontiguousarray(grids, dtype=np.uint8)
            if g.ndim != 3:
                raise ValueError("expected a (batch, height, width) stack")
            b, h, w = g.shape
            out = np.empty_like(g) if out is None else _check_out(out, g, "decode_batch")
            ctx = self._ctx or _ffi.default_context(0)
            _ffi.check(_ffi.lib().hgi_decode_u8_batch(ctx.handle, g.ctypes.data, w, h, int(levels), self._interp,
                                                      out.ctypes.data, b, h * w))
            return out
        import torch
        ctx = _torch_ctx(grids, self._ctx)
        if grids.dim() != 3:
            raise ValueError("expected a (batch, height, width) stack")
        b, h, w = grids.shape
        out = torch.empty_like(grids) if out is None else _check_out(out, grids, "decode_batch")
        _ffi.check(_ffi.lib().hgi_decode_u8_dev(ctx.handle, grids.data_ptr(), w, h, int(levels),
                                                self._interp, out.data_ptr(), b, h * w))
        return out
