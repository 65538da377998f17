"""Reference import path avgen/data/utils.py: the audio feature extractor (:26-110) and the file / dataset loaders
(:118-470), implemented in asva_amd."""
from asva_amd.audio_features import AudioMelspectrogramExtractor, waveform_to_melspectrogram  # noqa: F401
from asva_amd.data_utils import (get_evaluation_data, load_and_transform_images_stable_diffusion,  # noqa: F401
                                 load_audio_clips_uniformly, load_av_clips_uniformly, load_image, load_video_clips_uniformly)
