"""Drop-in import shim: the reference's public import paths (`avgen.pipelines.pipeline_audio_cond_animation`,
`avgen.models.unets`) resolved to the MI355X-native implementation in `asva_amd`, so the reference's
`scripts/animation_gen.py` (`from avgen.pipelines.pipeline_audio_cond_animation import generate_videos_for_dataset`,
scripts/animation_gen.py:4) runs unchanged with this repository first on PYTHONPATH."""
