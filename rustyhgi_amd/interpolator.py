"""Mirror of `hgi::interpolator` (reference src/interpolator.rs).

The reference's per-pixel trait method cannot cross to a GPU; the zero-sized
interpolator types select a device predictor instead.  A type
without a device predictor is rejected (HGI_EUNSUPPORTED), never run on the host.
"""
import enum


class InterpolationType(enum.IntEnum):   # src/interpolator.rs:4-9, metadata tag only
    Crossed = 0
    Line = 1
    Previous = 2


class Interpolator:                       # src/interpolator.rs:11-13
    kernel_id = None                      # hgi_interp value of the device predictor


class LeftTop(Interpolator):              # src/interpolator.rs:15-28
    kernel_id = 0


class Crossed(Interpolator):              # src/interpolator.rs:30-91
    kernel_id = 1
