"""MI355X-native GRU forward / BPTT behind the torch.nn.GRU interface, for TimeGAN's modules."""
from ._lib import (TgruError, lib, LAYOUT_TBF, LAYOUT_BTF, LAYOUT_PACKED,  # noqa: F401
                   reload_switches, launch_count, persist_timeouts, check_persist_timeouts)
from .functional import GRURunner, gru_forward_backward_raw, gru_apply, input_projection  # noqa: F401
from .gru import GRU  # noqa: F401
from .modules import (Embedding, Recovery, Supervisor, Generator, Discriminator,  # noqa: F401
                      build_networks, make_cfg)
from . import data, losses, parallel, scorer  # noqa: F401
from .head import head_apply  # noqa: F401
from .trainer import TimeGANTrainer  # noqa: F401
