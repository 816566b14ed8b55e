"""MI355X-native mel-synthesis path of Matcha-TTS-24k (text encoder + duration predictor + CFM decoder).

The directory name contains hyphens, so import it with
``importlib.import_module("matcha-tts-24k_amd")``; everything inside uses
relative imports.
"""
__version__ = "0.1.0"
