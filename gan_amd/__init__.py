"""gan_amd - MI355X-native lab3 codec-GAN training step (HIP kernels behind the reference's module API).

Import surface mirrors the reference's `lab 3/src` modules for this path:
    gan_amd.lab3_codec_models   CodecLatentTranslator, MultiScaleWaveDiscriminator, hinge / FM losses, CodecTrainWeights
    gan_amd.lab3_codec_train    CodecStageTrainConfig, train_codec_stage, _waveform_mrstft_loss, bank builders
    gan_amd.lab3_codec_bridge   FrozenEncodec.decode_embeddings (SEANet decoder)
The HIP library (gan_amd/libgan_amd.so, C ABI in include/gan_amd.h) is loaded on first kernel call; there is no
CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["lab3_codec_models", "lab3_codec_train", "lab3_codec_bridge", "ops", "optim", "dp"]
