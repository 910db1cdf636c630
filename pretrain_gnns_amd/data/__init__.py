from .batch import Data  # noqa: F401
from . import synthetic  # noqa: F401
from . import resident  # noqa: F401
