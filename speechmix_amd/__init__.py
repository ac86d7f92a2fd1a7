"""speechmix_amd: MI355X-native implementation of the SpeechMix fused training step."""
