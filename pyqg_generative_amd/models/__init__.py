from .parameterization import Parameterization
from .cgan_regression import CGANRegression
from .cvae_regression import CVAERegression
from .mean_var_model import MeanVarModel

__all__ = ['Parameterization', 'CGANRegression', 'CVAERegression', 'MeanVarModel']
