"""The handful of reference constants the hot path uses
(``playaid/constants.py:11,23,51`` and the defaults of
``playaid/ai_runner.py:426-437``)."""
import os

REPO_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SAVED_MODELS = os.path.join(REPO_ROOT, "models")
SAVED_ACTION_MODELS = os.path.join(SAVED_MODELS, "action")
AI_CACHE = os.path.join(REPO_ROOT, "ai_cache")

# class id == index; the label files carry this id (ai_runner.py:53-71)
CHAR_LIST = ["Byleth", "Diddy Kong", "Pikachu", "Joker", "Donkey Kong", "Jigglypuff"]

# hot-path geometry (ai_runner.py:432-439, 443, 417-418)
NUM_FRAMES_PER_SAMPLE = 7
FRAME_DELTA = 3
CROP_SIZE = 128
CROP_PADDING = 30
RESNET_FEATURE_SIZE = 1000
