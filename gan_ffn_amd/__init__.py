"""gan_ffn_amd — MI355X-native GAN-FFN generator/discriminator training step.

    from gan_ffn_amd import model      # drop-in for the reference's model.py hot-path classes
    from gan_ffn_amd import engine     # fast step runner (train_disc / train_gen / train_GAN counterpart)

The HIP library (gan_ffn_amd/lib/libganffn.so, built by __graft_entry__.build() or
`make -C gan_ffn_amd/csrc`) is required; nothing here falls back to CPU arithmetic.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "model", "engine", "data"]
