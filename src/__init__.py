"""Import root of the reference's own tests (``from src.teamoflow.mf.matrix_factorization import MatrixFactorization``,
/root/reference/test/test_loss.py:5-7, test_utils.py:5, test_predict.py:5-8): ``src.teamoflow`` is the ``teamoflow``
alias package, i.e. the MI355X engine in ``teamoflow_amd``."""
import sys

import teamoflow

sys.modules[__name__ + '.teamoflow'] = teamoflow
for _name, _mod in list(sys.modules.items()):
    if _name.startswith('teamoflow.'):
        sys.modules[__name__ + '.' + _name] = _mod
