"""MI355X-native attack inner loop behind SpeakerGuard's ``attack.*`` / ``model.*`` surface.

Importing the package does not load the HIP library; constructing a model or calling an op does,
and raises ``speakerguard_amd._native.NativeError`` when ``libspeakerguard_hip.so`` is missing.
"""
__version__ = "0.1.0"
