// Tuning constants and test switches of the library, in one place.
//
// The RELEASE library (make -> libhgi_hip.so) fixes every one of them at compile time: HGI_KNOB(NAME, DEFAULT) is DEFAULT,
// HGI_SWITCH(NAME) is false, no name string reaches the object file and nothing here touches the environment.  (The one
// variable the release library does read, HGI_NO_PLACEMENT, is part of the documented interface: include/hgi.h,
// hgi_planes_alloc.)
//
// The KNOBS build (make VARIANT=_knobs EXTRA=-DHGI_KNOBS_ENV -> libhgi_hip_knobs.so) takes each of them from the environment
// variable of the same name, read once.  It exists for two users, both of which load it explicitly through the Python
// binding's HGI_LIB_PATH: the GPU test suite, which forces every code path a shipped configuration can reach -- tile
// geometry, the byte-checked path, the host recursion for large lattice planes, un-banded host calls, the band-hold hook --
// on shapes small enough for the oracle; and the sweep tools under tools/, which vary one constant per process.
#pragma once

#ifdef HGI_KNOBS_ENV
#include <stdlib.h>
namespace hgi {
inline int knob_env(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
}  // namespace hgi
#define HGI_KNOB(NAME, DEFAULT)                                         \
    ([]() -> int {                                                      \
        static const int v_ = ::hgi::knob_env(#NAME, (int)(DEFAULT));   \
        return v_;                                                      \
    }())
#define HGI_SWITCH(NAME)                                  \
    ([]() -> bool {                                       \
        static const bool v_ = getenv(#NAME) != nullptr;  \
        return v_;                                        \
    }())
#else
#define HGI_KNOB(NAME, DEFAULT) ((int)(DEFAULT))
#define HGI_SWITCH(NAME) (false)
#endif
