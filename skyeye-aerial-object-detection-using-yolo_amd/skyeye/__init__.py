"""SkyEye detection forward pass on AMD Instinct MI355X (gfx950) -- drop-in for the reference's ``skyeye`` package
on the inference hot path (detector classes + non_max_suppression).  Everything computes in libskyeye_hip.so."""
__version__ = "0.1.0"
