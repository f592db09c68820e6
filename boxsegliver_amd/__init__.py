"""boxsegliver_amd -- MI355X-native U-Net segmentation hot path (HIP kernels behind a C ABI) with the
plugin surface of Jarvis73/BoxSegLiver (--model/--classes, BaseNet contract, CustomEstimator)."""
__version__ = "0.1.0"
