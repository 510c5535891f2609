"""Scoring graphs (plug-in interface of /root/reference/src/teamoflow/mf/predict_graphs.py).
The reference defines these but never calls them (matrix_factorization.py:149,195 hard-code the
dot product); kept for surface parity."""
from abc import ABC, abstractmethod

from .. import _ops


class PredictionGraph(ABC):
    """predict_graphs.py:6-21."""

    @abstractmethod
    def get_prediction(self, user_embedding, item_embedding):
        pass


class DotProductPrediction(PredictionGraph):
    """predict_graphs.py:24-35: user_embedding @ item_embedding^T on the fp32 MFMA."""

    def get_prediction(self, user_embedding, item_embedding):
        return _ops.predict_gemm(user_embedding, item_embedding)
