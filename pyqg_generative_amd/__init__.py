"""MI355X-native online parameterized-QG ensemble engine (drop-in for the
pyqg_generative online-stepping path).  See DESIGN.md / INTEGRATION.md."""
from . import _lib                      # raises ImportError if libqgx.so is not built
from .engine import EnsembleEngine, Generator

__all__ = ['EnsembleEngine', 'Generator']
