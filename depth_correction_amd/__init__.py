"""depth_correction_amd: the map-consistency training hot path of ctu-vras/depth_correction, MI355X-native.

Sub-modules mirror the reference package (depth_cloud, nearest_neighbors, utils, model, loss, filters, preproc,
eval, transform, config, train); ``ops`` / ``plan`` / ``pipeline`` are the array-level layers over the HIP library
(``lib/libdc_hip.so``, C ABI in ``include/dc_hip.h``).  Importing the package does not load the library; the first
kernel call does, and raises if it is missing -- there is no CPU implementation.
"""
__version__ = '0.1.0'


def install_as(alias='depth_correction'):
    """Register this package (and its sub-modules) under ``alias`` so that unmodified callers of the reference
    (``from depth_correction.depth_cloud import DepthCloud`` ...) import the MI355X-native implementation."""
    import importlib
    import sys
    pkg = sys.modules[__name__]
    sys.modules[alias] = pkg
    for name in ('config', 'dataset', 'depth_cloud', 'eval', 'filters', 'io', 'loss', 'metrics', 'model', 'nearest_neighbors',
                 'preproc', 'train', 'transform', 'utils'):
        sys.modules['%s.%s' % (alias, name)] = importlib.import_module('%s.%s' % (__name__, name))
    return pkg
