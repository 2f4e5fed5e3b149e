from . import models  # noqa: F401
from . import evaluation  # noqa: F401
