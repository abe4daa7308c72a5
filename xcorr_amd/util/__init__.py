"""xcorr_amd.util -- mirror of xcorr/util for the hot path (cc, hasher)."""
from . import cc  # noqa: F401
from . import hasher  # noqa: F401
