"""The ISA lint lives in the package (voxvae/isa_lint.py: the build runs it); this name is kept for the tests and scripts that import it."""
from voxvae.isa_lint import *  # noqa: F401,F403
from voxvae.isa_lint import PINNED_WAITS, NO_SCRATCH, RESIDENT_BUDGET, family, check_file, check_all  # noqa: F401
