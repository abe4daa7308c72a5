"""xcorr_amd.core -- the pair x window driver of xcorr/core (init, process)."""
from .init import init, CCDataset  # noqa: F401
from .process import process, process_batched  # noqa: F401
from .client import ArrayClient  # noqa: F401
from .postprocess import postprocess, postprocess_rows  # noqa: F401
from . import lazy  # noqa: F401
from . import timelapse  # noqa: F401

__all__ = ['init', 'process', 'process_batched', 'postprocess', 'postprocess_rows', 'lazy', 'timelapse', 'CCDataset', 'ArrayClient']
