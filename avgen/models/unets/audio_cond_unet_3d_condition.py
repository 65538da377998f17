from asva_amd.unet import AudioUNet3DConditionModel, UNet3DConditionOutput  # noqa: F401
