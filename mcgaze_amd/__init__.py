"""mcgaze_amd -- MI355X (gfx950) implementation of the MCGaze per-clip forward path behind the
reference's mmdet-style registry / python-config surface.  See DESIGN.md and INTEGRATION.md."""
from . import detector as _detector  # noqa: F401  (registers the model classes)
from .apis import init_detector, load_checkpoint
from .config import Config, ConfigDict
from .registry import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, DETECTORS, HEADS, LOSSES, MODELS, NECKS, BACKBONES,
                       ROI_EXTRACTORS, TRANSFORMER, Registry, build_backbone, build_detector, build_head, build_loss,
                       build_neck, build_roi_extractor, build_transformer)

__all__ = ['init_detector', 'load_checkpoint', 'Config', 'ConfigDict', 'Registry', 'MODELS', 'BACKBONES', 'NECKS', 'HEADS',
           'ROI_EXTRACTORS', 'LOSSES', 'DETECTORS', 'TRANSFORMER', 'BBOX_ASSIGNERS', 'BBOX_SAMPLERS', 'BBOX_CODERS',
           'build_backbone', 'build_neck', 'build_head', 'build_roi_extractor', 'build_loss', 'build_detector', 'build_transformer']
__version__ = '0.1.0'
