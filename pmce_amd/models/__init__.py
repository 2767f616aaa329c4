"""Host-side mirror of the reference's ``lib/models`` package for the inference hot path:
``models.PMCE.get_model``, ``models.PoseEstimation.get_model``, ``models.CoevoDecoder.get_model``
(reference lib/models/__init__.py:1-3).  Same constructor arguments, forward signatures, return order and
checkpoint layout; the arithmetic runs in libpmce_hip.so."""
from . import CoevoDecoder, PMCE, PoseEstimation  # noqa: F401
