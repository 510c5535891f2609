from . import matrix_factorization
from . import loss_graphs
from . import predict_graphs
from . import embedding_graphs
from . import initializer_graphs
from . import input_utils
from . import utils
from . import sparse

__all__ = ['matrix_factorization', 'loss_graphs', 'predict_graphs', 'embedding_graphs', 'initializer_graphs',
           'input_utils', 'utils', 'sparse']
