"""speechmix_amd: MI355X-native implementation of the SpeechMix fused training step.

The classes mirror what `from speechmix import *` gives the reference's train.py (ref:speechmix/__init__.py,
ref:train.py:190-225); they are imported lazily so that `import speechmix_amd` stays light (no kernels are loaded until a
model is built).
"""
__all__ = ["SpeechMixEED", "SpeechMixFixed", "SpeechMixAdapter", "SpeechMixSelf", "HFSpeechMixEED", "HFSpeechMixFixed",
           "HFSpeechMixAdapter", "HFSpeechMixSelf", "shift_tokens_right", "handle_decoder_input_none"]


def __getattr__(name):
    if name in __all__:
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
