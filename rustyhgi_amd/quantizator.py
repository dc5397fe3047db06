"""Mirror of `hgi::quantizator` (reference src/quantizator.rs).

A quantizer is a pure u8 -> u8 map, so any `Quantizator` crosses to the GPU as its
256-entry table; `Linear` builds its table through the C ABI.
"""
import ctypes
import enum

import numpy as np

from . import _ffi


class QuantizationLevel(enum.IntEnum):   # src/quantizator.rs:3-8 (bincode variant index)
    Lossless = 0
    Low = 1
    Medium = 2
    High = 3

    @classmethod
    def parse(cls, text):
        """Case-insensitive, as clap's `case_insensitive` (src/options.rs:61)."""
        for m in cls:
            if m.name.lower() == str(text).lower():
                return m
        raise ValueError("invalid quantization level %r (expected one of %s)" %
                         (text, ", ".join(m.name for m in cls)))


class Quantizator:                        # src/quantizator.rs:12-15
    def quantize(self, value):
        raise NotImplementedError

    def error(self):
        raise NotImplementedError

    def table(self):
        """The 256-entry tabulation handed to the device."""
        return np.array([self.quantize(i) & 0xFF for i in range(256)], dtype=np.uint8)


class NoOp(Quantizator):                  # src/quantizator.rs:17-34
    def __init__(self, _level=None):
        pass

    @classmethod
    def from_level(cls, _level):
        return cls()

    def quantize(self, value):
        return int(value) & 0xFF

    def error(self):
        return 0

    def table(self):
        t = np.empty(256, np.uint8)
        _ffi.lib().hgi_noop_lut(t.ctypes.data_as(ctypes.c_void_p))
        return t


class Linear(Quantizator):                # src/quantizator.rs:36-74
    def __init__(self, level):
        self.level = QuantizationLevel(level)
        self._table = np.empty(256, np.uint8)
        err = ctypes.c_uint8(0)
        _ffi.check(_ffi.lib().hgi_linear_lut(int(self.level), self._table.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.byref(err)))
        self._error = int(err.value)

    @classmethod
    def from_level(cls, level):           # `impl From<QuantizationLevel> for Linear`
        return cls(level)

    def quantize(self, value):            # :66-69
        return int(self._table[int(value) & 0xFF])

    def error(self):                      # :71-73
        return self._error

    def table(self):
        return self._table.copy()
