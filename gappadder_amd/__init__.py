"""gappadder_amd — MI355X-native read recruitment + per-gap local assembly (GAPPadder's hot path).

`gappadder_amd.hip_api.GapFill` is the object over the C ABI (include/gapfill_hip.h, libgapfill_hip.so);
the modules named after the reference's own files mirror its stage interfaces on top of it."""
__all__ = ["hip_api", "_lib"]
