"""MI355X-native hot path of aind-exaspim-image-compression (BM4D denoise, intensity transforms,
tiled BM4DNet inference) behind the reference's own operator API.

Importing this package needs neither a GPU nor torch; computing anything needs the built
``csrc/libexabm4d.so`` and an MI355X (there is no CPU fallback)."""

__version__ = "0.1.0"
