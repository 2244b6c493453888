"""depth_correction_amd: the map-consistency training hot path of ctu-vras/depth_correction, MI355X-native.

Sub-modules mirror the reference package (depth_cloud, nearest_neighbors, utils, model, loss, filters, preproc,
eval, transform, config, train); ``ops`` / ``plan`` / ``pipeline`` are the array-level layers over the HIP library
(``lib/libdc_hip.so``, C ABI in ``include/dc_hip.h``).  Importing the package does not load the library; the first
kernel call does, and raises if it is missing -- there is no CPU implementation.
"""
__version__ = '0.1.0'
