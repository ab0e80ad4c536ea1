"""nero_amd: MI355X-native NeRO Stage-I/II render step (HIP/CDNA4 kernels behind the reference's renderer API)."""
__version__ = '0.1.0'
