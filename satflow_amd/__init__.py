"""satflow_amd: MI355X-native hot path of openclimatefix/satflow (ConvLSTM encoder-decoder and the
MetNet-style stack) behind the reference's model surface.  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

from ._hip import check_device_errors, clear_device_errors, compute_dtype_name, device_errors, set_compute_dtype  # noqa: E402,F401
