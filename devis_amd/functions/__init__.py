# mirrors /root/reference/src/models/ops/functions/__init__.py:9 (same exported names)
from .ms_deform_attn_func import (MSDeformAttnFunction, MSDeformAttnTemporalFunction,  # noqa: F401
                                  ms_deform_attn_core_pytorch, project_value,
                                  MSDeformPrepFunction, MSDeformPrepFusedFunction)
