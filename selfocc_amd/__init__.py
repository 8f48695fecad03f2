"""selfocc_amd — MI355X-native implementation of SelfOcc's data-parallel hot path
(TPV/BEV lifter MSDA + SDF volume-rendering head + reprojection loss) behind the
reference's registry / config surface.  The arithmetic lives in csrc/*.hip behind the C ABI
of include/selfocc_hip.h; there is no CPU fallback."""
__version__ = "0.1.0"
