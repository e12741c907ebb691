"""Flat hyper-parameter namespace, drop-in for the reference's ``settings.py``.

``from settings import *`` publishes the same names the reference's scripts star-import
(reference settings.py:1-245).  Values come from ``midi_vae_amd.config.build_settings()``; to run a
different configuration set the environment variable ``MIDIVAE_SETTINGS`` to a JSON object of base knobs
(e.g. ``{"cell_type": "LSTM", "latent_dim": 64, "input_length": 128, "output_length": 128}``) before import,
or call ``build_settings(**knobs)`` directly.  Unlike the reference, importing this module creates no
directories (reference settings.py:58-61).
"""
import json as _json
import math  # noqa: F401  (the reference's scripts receive math / np / os / time through this star-import)
import os
import time  # noqa: F401

import numpy as np  # noqa: F401

import midi_vae_amd  # noqa: F401
from midi_vae_amd.config import build_settings as _build

_knobs = _json.loads(os.environ.get("MIDIVAE_SETTINGS", "{}"))
globals().update(_build(**_knobs))
