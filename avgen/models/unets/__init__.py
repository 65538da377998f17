from asva_amd.unet import AudioUNet3DConditionModel  # noqa: F401  (reference: avgen/models/unets/__init__.py:1)
