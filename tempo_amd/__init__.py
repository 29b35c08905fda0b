"""tempo_amd: MI355X-native heatmap -> voxel unprojection for TEMPO.

Drop-in replacements for the reference's ``tempo/models/project_layer.py``
modules, running hand-written HIP kernels (gfx950) through a C ABI.
"""
from .project_layer import (PackedHeatmaps, ProcessLayerNHWC, ProjectLayer,  # noqa: F401
                            ProjectLayerWithMask)

__all__ = ['ProjectLayer', 'ProjectLayerWithMask', 'ProcessLayerNHWC', 'PackedHeatmaps']
