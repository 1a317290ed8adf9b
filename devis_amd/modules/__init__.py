# mirrors /root/reference/src/models/ops/modules/__init__.py:9 (same exported names)
from .ms_deform_attn import (MSDeformAttn, TemporalMSDeformAttnBase,  # noqa: F401
                             TemporalMSDeformAttnDecoder, TemporalMSDeformAttnEncoder)
