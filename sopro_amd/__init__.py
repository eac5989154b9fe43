"""sopro_amd: MI355X-native engine for the Sopro TTS synthesize/stream hot path.

``SoproTTS`` is the only public name, as in the reference package (src/sopro/__init__.py:3-5).
Importing the package does not need a GPU; constructing an engine does (and raises otherwise).
"""
from .tts import SoproTTS  # noqa: F401

__all__ = ["SoproTTS"]
