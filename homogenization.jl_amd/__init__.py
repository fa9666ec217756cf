"""homogenization.jl_amd: MI355X-native hot path (matrix-free multigrid on the implicit fine grid) of
haampie/Homogenization.jl behind the reference's own interface.  See DESIGN.md / INTEGRATION.md."""
from . import _lib  # noqa: F401
from .api import *  # noqa: F401,F403
from . import api  # noqa: F401
from . import driver  # noqa: F401
from . import dist  # noqa: F401
