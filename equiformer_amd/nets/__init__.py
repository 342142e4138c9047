"""Drop-in replacement of the reference's `nets` package for the graph-attention-transformer hot path:
same registry (`nets.model_entrypoint(name)`), same factory signatures, same module tree / parameter names,
forward and backward executed by libequiformer_hip.so on MI355X."""
from .registry import model_entrypoint, register_model, list_models  # noqa: F401

from .graph_attention_transformer import *  # noqa: F401,F403
from .graph_attention_transformer_md17 import *  # noqa: F401,F403
from .graph_attention_transformer_oc20 import *  # noqa: F401,F403
from .dp_attention_transformer import *  # noqa: F401,F403
from .equiformer_md17_dens import *  # noqa: F401,F403
