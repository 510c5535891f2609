"""Alias package: ``from teamoflow.mf import matrix_factorization`` (README.md:112-119 of the
reference) resolves to the MI355X engine in ``teamoflow_amd``."""
import sys

import teamoflow_amd
from teamoflow_amd import mf

__version__ = teamoflow_amd.__version__

sys.modules[__name__ + '.mf'] = mf
for _name in mf.__all__:
    sys.modules[f'{__name__}.mf.{_name}'] = getattr(mf, _name)
