"""asr_chinese_e2e_amd: MI355X-native training hot path of the zqs01/ASR_chinese_e2e Speech-Transformer.

Importing the package does not touch the GPU; `asr_chinese_e2e_amd.kernels` loads libasr_hip.so
(no CPU fallback - see _lib.py)."""
__version__ = "0.1.0"
