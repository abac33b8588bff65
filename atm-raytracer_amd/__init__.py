"""MI355X-native per-pixel ray-marching path of atm-raytracer (host-side mirror of the reference's
generator interface above the C ABI in include/atmrt.h).  See DESIGN.md."""
__version__ = "0.1.0"
