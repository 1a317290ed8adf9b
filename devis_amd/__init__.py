"""devis_amd -- MI355X-native (gfx950 / CDNA4) multi-scale deformable attention for DeVIS.

A from-scratch replacement for DeVIS's only native component, ``src/models/ops`` (the vendored
Deformable-DETR CUDA extension ``MultiScaleDeformableAttention`` plus its Python wrappers), behind the
same Python surface::

    from devis_amd.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch
    from devis_amd.modules import MSDeformAttn, TemporalMSDeformAttnEncoder, TemporalMSDeformAttnDecoder

Arithmetic lives in hand-written HIP kernels behind the C ABI of ``include/msda.h``
(``devis_amd/csrc/*.hip``, one translation unit per kernel family -> ``devis_amd/libmsda_hip.so``); the Python here is the host side.
There is no CPU fallback: like the reference (``src/ms_deform_attn.h:38,60``) the operator raises on
CPU tensors, and it raises if the HIP library cannot be loaded.
"""
from .functions import (MSDeformAttnFunction, MSDeformAttnTemporalFunction,  # noqa: F401
                        ms_deform_attn_core_pytorch)
from .modules import (MSDeformAttn, TemporalMSDeformAttnDecoder,  # noqa: F401
                      TemporalMSDeformAttnEncoder)
from .argument_builders import patch_transformer  # noqa: F401
from .graphs import graphed, graph_stream, GraphedLayer  # noqa: F401
from .tuning import tune  # noqa: F401

__all__ = ["MSDeformAttnFunction", "MSDeformAttnTemporalFunction", "ms_deform_attn_core_pytorch",
           "MSDeformAttn", "TemporalMSDeformAttnEncoder", "TemporalMSDeformAttnDecoder", "patch_transformer", "graphed", "graph_stream", "GraphedLayer", "tune"]
