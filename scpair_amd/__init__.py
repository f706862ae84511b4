"""scpair_amd -- MI355X-native implementation of scPair's training hot path.

The public namespace is the flat union of the modules, like the reference's scpair/__init__.py:1-9.
"""
from .scVI_distribution import (NegativeBinomial, ZeroInflatedNegativeBinomial, log_nb_positive,  # noqa: F401
                                log_zinb_positive)
from .loss import binarized_loss, raw_loss, scaled_loss  # noqa: F401
from .utils import (FCNN, HipFCNN, HipLinear, add_weight_decay, eval_net, eval_VAE, evaluating_mode,  # noqa: F401
                    init_weights, set_seed, train_net, train_VAE, training_mode)
from .model import (Input_Module, LazyGPUDataset, Module_Module, Output_Module_Ber, Output_Module_Gau,  # noqa: F401
                    Output_Module_Raw, clear_gpu_memory, move_to_device_safe, scPair_object)
from .pairedVAE import VAE_DECODER, VAE_ENCODER  # noqa: F401
from .vae_trainer import scVAE_object  # noqa: F401
from .sparse import BatchTiles, DeviceCSR  # noqa: F401
from .ops import FusedAdamW, get_precision, set_precision  # noqa: F401
from .evaluator import FOSCTTM, FOSFTTM  # noqa: F401
from .data import check_distribution, merge_paired_data, split_data, training_split  # noqa: F401

__version__ = "0.1.0"
