"""smrender_amd -- MI355X-native back end for smrender's projection / draw / caption-scan path.

The compute lives in ``libsmr_hip.so`` (hand-written HIP for gfx950, C-ABI in
``include/smr_hip.h``).  This package is the thin Python host layer used by the tests and
``bench.py``; the production binding is the C plugin layer inside the same library
(``act_draw_*`` / ``act_cap_*``).  There is no CPU fallback: importing
works anywhere, but every compute call raises if the library or a GPU is missing.
"""
from .hip import (SmrHipError, lib, init, shutdown, DeviceArray, Projection, DrawStyle, Canvas,
                  project, unproject, poly_area, Objects, set_projection, timings, set_profiling, sync, device_name,
                  caprot_steps)

__all__ = ["SmrHipError", "lib", "init", "shutdown", "DeviceArray", "Projection", "DrawStyle", "Canvas",
           "project", "unproject", "poly_area", "Objects", "set_projection", "timings", "set_profiling", "sync", "device_name",
           "caprot_steps"]
