"""xcorr_amd.util -- mirror of xcorr/util for the hot path (cc only)."""
from . import cc  # noqa: F401
