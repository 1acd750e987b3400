"""visgeom_amd -- MI355X-native reprojection residual / Jacobian engine for visgeom's calibration
hot path (GenericProjectionJac::Evaluate + the normal-equation build).  HIP-only: importing the
sub-modules that compute needs visgeom_amd/lib/libvisgeom_amd.so (see __graft_entry__.build())."""
__version__ = "0.1.0"

from . import capi  # noqa: F401
from .problem import BlockGroup, CalibrationProblem, GenericProjectionJac  # noqa: F401


def release_cached_memory():
    """hand the solver's cached device / pinned work blocks back to the driver (vg_release_cached_memory)"""
    capi.load().vg_release_cached_memory()
