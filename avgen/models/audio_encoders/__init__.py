from asva_amd.audio_encoder import ImageBindSegmaskAudioEncoder  # noqa: F401  (reference: avgen/models/audio_encoders/__init__.py)
