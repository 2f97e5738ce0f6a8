// The library's build id: first 16 hex digits of the sha256 over the kernel sources and the public header (Makefile: BUILD_ID;
// mcgaze_amd/lib.py::source_id is the same recipe).  Its own translation unit, rebuilt whenever any of those files changes: compiled into
// an object that only SOME source changes rebuild, the id went stale (round 5: decoder.hip changed, the id still named the previous build).
#include "../../include/mcgaze_hip.h"
#ifndef MCG_BUILD_ID
#define MCG_BUILD_ID "unknown"
#endif
extern "C" const char* mcg_build_id(void) { return MCG_BUILD_ID; }
