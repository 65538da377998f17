"""Reference import path of the audio feature extractor (avgen/data/utils.py:26-110); file / video loaders are out of scope."""
from asva_amd.audio_features import AudioMelspectrogramExtractor, waveform_to_melspectrogram  # noqa: F401
