"""qsparse_amd -- MI355X-native quantize/prune operators with the API of mlzxy/qsparse v2.0.1.

Drop-in for the reference's training hot path: ``quantize()``, ``prune()``, ``convert()`` and the layers,
callbacks and helpers they expose (reference qsparse/__init__.py:2-8).  GPU tensors are processed by
hand-written HIP kernels for gfx950 behind a C ABI (``include/qsparse_hip.h``,
``qsparse_amd/libqsparse_hip.so``); see DESIGN.md.
"""
from qsparse_amd.batch import WeightBatcher
from qsparse_amd.convert import convert
from qsparse_amd.export import LayerExport, QuantizedTensor, export_integer
from qsparse_amd.fuse import fuse_bn
from qsparse_amd.graphs import resync_host_state
from qsparse_amd.quantize import (AdaptiveQuantizer, DecimalQuantizer, ScalerQuantizer, quantize,
                                  quantize_with_decimal, quantize_with_line, quantize_with_scaler)
from qsparse_amd.sparse import (MagnitudePruningCallback, UniformPruningCallback, devise_layerwise_pruning_schedule,
                                prune)
from qsparse_amd.util import (auto_name_prune_quantize_layers, calculate_mask_given_importance, extra_state_dict,
                              load_extra_state_dict, preload_qsparse_state_dict)
from qsparse_amd.util import get_option as get_qsparse_option
from qsparse_amd.util import set_options as set_qsparse_options

from qsparse_amd.util import PINNED_TORCH, check_torch_pin

__version__ = "2.0.1+mi355x.1"
check_torch_pin()      # warns once under a torch other than the one the staged means' summation order was taken from
