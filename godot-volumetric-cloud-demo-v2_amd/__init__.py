"""MI355X-native volumetric cloud sky: host-side mirror of the reference's GDScript drivers
(cloud_sky/cloud_sky.gd, sky_lut.gd, transmittance_lut.gd) over the C ABI of libcloudsky.so
(include/cloudsky.h).  The HIP library is the product; this package is the thin caller.  There is no CPU
render path: constructing a renderer without the library or without a GPU raises."""
from . import _lib, assets, tiling  # noqa: F401
from ._lib import CloudSkyError, Context, MultiContext, lib, library_path  # noqa: F401
from .cloud_sky import CloudSky, FrameData  # noqa: F401
from .sky_lut import SkyLut  # noqa: F401
from .transmittance_lut import TransmittanceLut  # noqa: F401

__all__ = ["CloudSky", "FrameData", "SkyLut", "TransmittanceLut", "Context", "MultiContext", "CloudSkyError", "assets", "tiling", "lib",
           "library_path"]
