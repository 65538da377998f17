"""Same module path as the reference's pipeline file; everything is implemented in asva_amd.pipeline."""
from asva_amd.pipeline import (AudioCondAnimationPipeline, generate_videos, generate_videos_for_dataset,  # noqa: F401
                               synthetic_clip)
from asva_amd.schedulers import DDIMScheduler, PNDMScheduler  # noqa: F401
