"""visgeom_amd -- MI355X-native reprojection residual / Jacobian engine for visgeom's calibration
hot path (GenericProjectionJac::Evaluate + the normal-equation build).  HIP-only: importing the
sub-modules that compute needs visgeom_amd/lib/libvisgeom_amd.so (see __graft_entry__.build())."""
__version__ = "0.1.0"

from . import capi  # noqa: F401
from .problem import BlockGroup, CalibrationProblem, GenericProjectionJac  # noqa: F401
