"""esrecsys_amd -- MI355X-native embedding-training hot path of BBischof/ESRecsys.

Only the data-parallel hot path is here: the GloVe co-occurrence model and the
Shop-The-Look two-tower score/loss head, their train steps, the sparse optimizer, brute-force
retrieval and row-shard routing.  Host code is Python (as the reference's is) over a C-ABI shared
library of hand-written gfx950 HIP kernels (include/esr_hip.h, esrecsys_amd/csrc/).

    from esrecsys_amd.wikipedia.models import Glove
    from esrecsys_amd.wikipedia.train_cooccurence import apply_model, update_model, find_knn
    from esrecsys_amd.pinterest.models import STLModel
    from esrecsys_amd.pinterest.train_shop_the_look import train_step, eval_step
    from esrecsys_amd.pinterest.make_recommendations import find_top_k
    from esrecsys_amd import optim, TrainState            # optax.adam / TrainState look-alikes
"""
from . import train_state as optim  # noqa: F401  (optim.adam, optim.sparse_adagrad, optim.sgd)
from ._lib import EsrLibraryError, LIB_PATH  # noqa: F401
from .train_state import RowGrads, SegmentIndex, TrainState  # noqa: F401

__version__ = "0.1.0"
