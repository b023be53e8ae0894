"""`from utils import *` surface of the reference (utils/__init__.py:1-7), restricted to the
retrieval path: metrics, batch fold, device/CLI helpers, dataset listing."""
from .general import *          # noqa: F401,F403
from .dataset import *          # noqa: F401,F403
from .metrics import *          # noqa: F401,F403
from .train_general import *    # noqa: F401,F403
from .train_siamese import *    # noqa: F401,F403
